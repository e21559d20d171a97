/*
 * sgtd_oracle.h — C interface of the CPU ORACLE.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference (Hfx-J/SGTD) ships no tests, golden vectors
 * or fixtures for this path and cannot be compiled in this image (it needs
 * Eigen, PCL/FLANN, ROS and Ceres headers, none of which are present), so
 * this restatement is pinned only by the hand-derived known-answer tests in
 * tests/test_oracle_kat.py.
 */
#ifndef SGTD_ORACLE_H
#define SGTD_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_config {
  int32_t descriptor_near_num;   /* K    — STDesc.cpp:179  */
  int32_t candidate_num;         /*      — STDesc.cpp:423  */
  int32_t max_frame_n;           /* MAX_FRAME_N, STDesc.h:33 (runtime here) */
  int32_t num_threads;           /* OpenMP threads for the parallel regions */
  double descriptor_min_len;     /*      — STDesc.cpp:181  */
  double descriptor_max_len;     /*      — STDesc.cpp:180  */
  double std_side_resolution;    /*      — STDesc.cpp:178  */
  double rough_dis_threshold;    /*      — STDesc.cpp:357  */
} orc_config;

/* Descriptor export layout (structure of arrays, caller allocated). */
typedef struct orc_desc_soa {
  double *side;      /* [n*3]  side_length_            */
  double *angle;     /* [n*3]  angle_                  */
  double *center;    /* [n*3]  center_                 */
  double *vertex;    /* [n*9]  vertex_A_,B_,C_ (xyz)   */
  int32_t *label;    /* [n*3]  vertex_attached_ as int */
  uint32_t *frame;   /* [n]    frame_id_               */
  int32_t *node_id;  /* [n*3]  node_id = {i,m,n}       */
} orc_desc_soa;

typedef struct orc_counters {
  int64_t D;   /* query descriptors handed to the last select     */
  int64_t P;   /* table entries visited by the inner loop (:372)   */
  int64_t M;   /* rough matches emitted (:378-384)                 */
  int64_t E;   /* entries in the table                             */
  int64_t U;   /* distinct keys (buckets) in the table             */
  double probe_ms;   /* the reference's CS1 span (:321..:403)      */
  double select_ms;  /* whole candidate_selector                   */
  double build_ms;   /* last BuildSingleScanSTD                    */
} orc_counters;

typedef struct orc_manager orc_manager;

orc_manager *orc_create(const orc_config *cfg);
void orc_destroy(orc_manager *m);

uint32_t orc_current_frame_id(const orc_manager *m);
/* start value of current_frame_id_ (a table shard of a multi-GPU map starts at its first frame) */
void orc_set_current_frame_id(orc_manager *m, uint32_t id);
/* OpenMP threads of the next select/verify calls (timing protocol) */
void orc_set_num_threads(orc_manager *m, int n);

/* Combinatorial_Binary_Encoding — STDesc.cpp:3-16 */
int orc_label_code(int a, int b, int c);
/* the restated key types: STDesc_LOC (x, y, z, a, b, c; STDesc.h:217-250) and VOXEL_LOC
 * (x, y, z; STDesc.h:126-154) — equality and std::hash value */
int orc_cell_key_eq(const int64_t *p, const int64_t *q);
int64_t orc_cell_key_hash(const int64_t *p);
int orc_milli_key_eq(const int64_t *p, const int64_t *q);
int64_t orc_milli_key_hash(const int64_t *p);

/* BuildSingleScanSTD (STDesc.cpp:174-315) on keypoints xyz[n*3] f32 + label[n].
 * The result is kept inside the manager ("last built"); returns its size. */
int64_t orc_build(orc_manager *m, const float *xyz, const uint32_t *label, int n);
/* copy the last built descriptors out (arrays sized by orc_build's return) */
void orc_last_export(const orc_manager *m, orc_desc_soa *out);
/* AddSTDescs (STDesc.cpp:149-172) of the last built descriptors */
void orc_add_last(orc_manager *m);
/* n_frames x (orc_build, orc_add_last) on uniform frames of n keypoints; builds in parallel, inserts in order */
void orc_add_frames(orc_manager *m, const float *xyz, const uint32_t *label, int n_frames, int n);
/* AddSTDescs of caller-provided descriptors */
void orc_add(orc_manager *m, const orc_desc_soa *d, int64_t n);

/* candidate_selector (STDesc.cpp:318-460).
 *   use_last != 0 : query = last built descriptors, else query = (q, nq).
 * Outputs (all optional except n_cand):
 *   cand_frame/cand_votes [candidate_num]
 *   cand_off  [candidate_num+1]  offsets into the per-candidate match arrays
 *   the per-candidate match lists are kept inside the manager and read with
 *   orc_cand_matches(); the full rough-match list with orc_rough_matches(). */
int orc_select(orc_manager *m, int use_last, const orc_desc_soa *q, int64_t nq,
               int32_t *cand_frame, int32_t *cand_votes, int64_t *cand_off,
               int32_t *n_cand);

/* total number of pairs over all candidates of the last select */
int64_t orc_cand_match_total(const orc_manager *m);
/* (query descriptor index, global insertion index of the table entry) pairs,
 * candidate after candidate, in match_list_ order (STDesc.cpp:437-449) */
void orc_cand_matches(const orc_manager *m, int32_t *q_idx, int64_t *db_entry);

/* the rough matches of the last select in (i, cell, j) order (:378-384):
 * q_idx, cell index 0..26 in voxel_round order, j, global entry id, frame, dis */
int64_t orc_rough_total(const orc_manager *m);
void orc_rough_matches(const orc_manager *m, int32_t *q_idx, int32_t *cell,
                       int32_t *j, int64_t *db_entry, uint32_t *frame,
                       double *dis);

/* votes histogram (match_array, :323,:410) of the last select, [max_frame_n] */
void orc_votes(const orc_manager *m, double *votes);

/* read table entries by global insertion index */
void orc_fetch_entries(const orc_manager *m, const int64_t *db_entry, int64_t n,
                       orc_desc_soa *out);

/* table dump in a canonical order (key ascending as (a,x,y,z), bucket order
 * inside): keys [U*4] = x,y,z,a ; bucket_off [U+1] ; entry ids [E] */
void orc_table_dump(const orc_manager *m, int64_t *keys, int64_t *bucket_off,
                    int64_t *entry_ids);

void orc_get_counters(const orc_manager *m, orc_counters *c);

/* "next" row: candidate_verify + triangle_solver
 * (STDesc.cpp:462-571) with a self-contained one-sided Jacobi SVD standing in
 * for Eigen::JacobiSVD (parity with Eigen unpinned).  cand = index into the
 * last select's candidate list.  Returns verify_score (-1 if rejected);
 * rot is row-major 3x3. n_success/success_idx list the kept pair indices. */
double orc_verify(orc_manager *m, int cand, double *t, double *rot,
                  int32_t *success_idx, int32_t *n_success);

#ifdef __cplusplus
}
#endif
#endif
