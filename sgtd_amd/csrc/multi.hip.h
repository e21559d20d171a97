// multi.hip.h — one handle, several GPUs: sgtd_create_multi gives a handle
// whose table is sharded over n_dev devices of ONE process.  Map frames are dealt to the
// devices in blocks of SGTD_SHARD_BLOCK consecutive frames, round robin; every device holds a
// complete table for its frames (in a dense local frame-id space), so a frame's vote count is
// final on its owner.  A query batch goes to every device (host pointers, each device stages
// its own copy and sweeps its shard concurrently — all launches are asynchronous), the
// per-device top-candidate_num tables (candidate_num * 8 B per query and device) come back to
// the host and are merged with the reference's rule (votes descending, ties -> lowest GLOBAL
// frame id, at least 5 votes, STDesc.cpp:423-433): the single-table candidate list, bit for
// bit, because the global top-k of disjoint frame sets is the top-k of the union of the local
// top-k lists.  Match lists, entries and geometric verification stay with the owner and are
// fetched from it per candidate.  Entry ids handed out by the group carry the owner in their
// upper bits.  (One process per GPU with RCCL: sgtd_amd/dist.py.)
#pragma once

#define SGTD_SHARD_BLOCK 64
#define SGTD_ENTRY_SHARD_SHIFT 40   // group entry id = shard << 40 | insertion index inside the shard

namespace multi {

struct Group {
  std::vector<sgtd_engine *> dev;        // one ordinary engine per device
  int n = 0;
  u32 current_frame_id = 0;              // global
  int nq = 0;
  bool batch_valid = false, merged = false, verified = false;
  // merged candidate tables of the last batch
  std::vector<int> n_cand, cand_frame, cand_votes, owner, owner_slot;
  std::vector<long long> pair_off;       // [nq * (cn + 1)] offsets into the query's concatenated lists
};

inline int shard_of(u32 g, int n) { return (int)((g / SGTD_SHARD_BLOCK) % (u32)n); }
inline u32 local_of(u32 g, int n) { return (g / (SGTD_SHARD_BLOCK * (u32)n)) * SGTD_SHARD_BLOCK + g % SGTD_SHARD_BLOCK; }
inline u32 global_of(int s, u32 l, int n) { return ((l / SGTD_SHARD_BLOCK) * (u32)n + (u32)s) * SGTD_SHARD_BLOCK + l % SGTD_SHARD_BLOCK; }

}  // namespace multi

namespace multi {
int create(const sgtd_config *cfg, const int *device_ids, int n_dev, sgtd_handle *out);
int destroy(sgtd_engine *e);
int build(sgtd_engine *e, const float *xyz, const uint32_t *label, int n, sgtd_desc_soa *out, int64_t capacity, int64_t *n_out);
int add(sgtd_engine *e, const sgtd_desc_soa *d, int64_t n);
int add_frames(sgtd_engine *e, const float *xyz, const uint32_t *label, const int64_t *kp_off, int n_frames, int device_ptrs);
int finalize(sgtd_engine *e);
int query_frames(sgtd_engine *e, const float *xyz, const uint32_t *label, const int64_t *kp_off, int n_queries, int device_ptrs);
int query_descs(sgtd_engine *e, const sgtd_desc_soa *q, int64_t nq);
int result_candidates(sgtd_engine *e, int32_t *n_cand, int32_t *cand_frame, int32_t *cand_votes, int64_t *pair_off);
int result_pairs(sgtd_engine *e, int q, int32_t *q_idx, int64_t *db_entry, int64_t capacity, int64_t *n_pairs);
int fetch_entries(sgtd_engine *e, const int64_t *db_entry, int64_t n, sgtd_desc_soa *out);
int result_query_desc_count(sgtd_engine *e, int q, int64_t *n);
int result_query_descs(sgtd_engine *e, int q, sgtd_desc_soa *out, int64_t capacity, int64_t *n_out);
int result_votes(sgtd_engine *e, int q, uint32_t *votes, int64_t capacity, uint32_t *frame_lo, int64_t *n);
int verify(sgtd_engine *e);
int result_verify(sgtd_engine *e, int q, double *score, double *pose);
int result_inliers(sgtd_engine *e, int q, int cand, int32_t *idx, int64_t capacity, int64_t *n);
int result_inlier_pairs(sgtd_engine *e, int q, int64_t *cand_off, int32_t *q_idx, int64_t *db_entry, int64_t capacity, int64_t *n_pairs);
int search_loop(sgtd_engine *e, double icp_threshold, int32_t *best_cand, int32_t *best_frame, double *best_score);
int sync(sgtd_engine *e);
int get_stats(sgtd_engine *e, sgtd_stats *out);
int max_batch(sgtd_engine *e, int n_keypoints, int64_t *max_queries);
u32 current_frame_id(sgtd_engine *e);
int device_count(sgtd_engine *e);
sgtd_engine *device_handle(sgtd_engine *e, int k);
}  // namespace multi
