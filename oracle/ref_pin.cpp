// ref_pin.cpp — TEST INFRASTRUCTURE.  The only pieces of the reference's hot path that compile
// with standard headers alone, compiled VERBATIM: the Makefile cuts them out of the reference's
// own files by line range into oracle/_ref/*.inc (never committed, never copied into the repo)
// and this wrapper gives them a C ABI, so that tests can pin the oracle's and the kernels'
// label code (row a7) and table-key equality / hash (rows a5, a8) against the reference's code
// itself:
//   consts.inc      src/sgtd/include/desc/STDesc.h:31-33    HASH_P, MAX_N, MAX_FRAME_N
//   cbe.inc         src/sgtd/src/STDesc.cpp:3-16            Combinatorial_Binary_Encoding
//   voxel_loc.inc   src/sgtd/include/desc/STDesc.h:126-136  class VOXEL_LOC (per-frame dedup key)
//   voxel_hash.inc  src/sgtd/include/desc/STDesc.h:147-154  std::hash<VOXEL_LOC>
//   stdesc_loc.inc  src/sgtd/include/desc/STDesc.h:217-250  class STDesc_LOC + std::hash (table key)
// Everything else of STDesc.h/.cpp needs Eigen, PCL, ROS or Ceres (STDesc.h:6-29) and cannot be
// built in this image; no stand-in headers are written.
#include <bitset>
#include <cstdint>
#include <functional>
#include <string>

#include "_ref/consts.inc"
#include "_ref/cbe.inc"
#include "_ref/voxel_loc.inc"
#include "_ref/voxel_hash.inc"
#include "_ref/stdesc_loc.inc"

extern "C" {
int ref_label_code(int a, int b, int c) { return Combinatorial_Binary_Encoding(a, b, c); }
int64_t ref_hash_p(void) { return HASH_P; }
int64_t ref_max_n(void) { return MAX_N; }
int64_t ref_max_frame_n(void) { return MAX_FRAME_N; }
int ref_voxel_eq(const int64_t *p, const int64_t *q) { return VOXEL_LOC(p[0], p[1], p[2]) == VOXEL_LOC(q[0], q[1], q[2]); }
int64_t ref_voxel_hash(const int64_t *p) { return std::hash<VOXEL_LOC>()(VOXEL_LOC(p[0], p[1], p[2])); }
// p, q: x, y, z, a, b, c
int ref_loc_eq(const int64_t *p, const int64_t *q) {
  return STDesc_LOC(p[0], p[1], p[2], p[3], p[4], p[5]) == STDesc_LOC(q[0], q[1], q[2], q[3], q[4], q[5]);
}
int64_t ref_loc_hash(const int64_t *p) { return std::hash<STDesc_LOC>()(STDesc_LOC(p[0], p[1], p[2], p[3], p[4], p[5])); }
}
