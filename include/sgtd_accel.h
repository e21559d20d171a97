/*
 * sgtd_accel.h — C ABI of the MI355X (gfx950) triangle-descriptor build +
 * geometric-hash match engine.  This is the drop-in boundary for the hot path
 * of Hfx-J/SGTD: the bodies of
 *
 *   STDescManager::BuildSingleScanSTD   src/sgtd/src/STDesc.cpp:174-315
 *   STDescManager::AddSTDescs           src/sgtd/src/STDesc.cpp:149-172
 *   STDescManager::candidate_selector   src/sgtd/src/STDesc.cpp:318-460
 *
 * (class declared at src/sgtd/include/desc/STDesc.h:342-440) forward to these
 * entry points; the reference has no FFI layer of its own, the C++ class *is*
 * the operator API.  INTEGRATION.md shows the adapter.
 *
 * Conventions: every function returns 0 (SGTD_OK) or a negative sgtd_status;
 * nothing throws across the ABI; all output buffers are caller allocated and
 * sized by the documented bounds; a handle is not thread safe (neither is the
 * reference's manager).  All HIP work of a handle is issued on one stream
 * (sgtd_set_stream).  There is NO CPU fallback: without a usable gfx950 device
 * sgtd_create fails with SGTD_ERR_NO_DEVICE.
 */
#ifndef SGTD_ACCEL_H
#define SGTD_ACCEL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum sgtd_status {
  SGTD_OK = 0,
  SGTD_ERR_INVALID = -1,      /* bad argument / config                         */
  SGTD_ERR_NO_DEVICE = -2,    /* no HIP device / wrong architecture            */
  SGTD_ERR_HIP = -3,          /* a HIP runtime call failed (see sgtd_last_error)*/
  SGTD_ERR_CAPACITY = -4,     /* caller buffer too small                       */
  SGTD_ERR_FRAME_LIMIT = -5,  /* frame id >= max_frame_n (MAX_FRAME_N, STDesc.h:33:
                                 the reference indexes a fixed array there)   */
  SGTD_ERR_UNSUPPORTED = -6,  /* configuration outside the kernels' envelope   */
  SGTD_ERR_STATE = -7,        /* call order (e.g. results before a query)      */
  SGTD_ERR_IO = -8            /* a graph file could not be opened or parsed    */
} sgtd_status;

/* ConfigSetting fields the path reads (STDesc.h:38-72); defaults in
 * sgtd_default_config() are the shipped YAML (config/SG_localization.yaml:74-89) */
typedef struct sgtd_config {
  int32_t descriptor_near_num;   /* K, STDesc.cpp:179; 3 <= K <= 16            */
  int32_t candidate_num;         /* STDesc.cpp:423; <= 64                      */
  int32_t max_frame_n;           /* MAX_FRAME_N, STDesc.h:33 (runtime here)    */
  int32_t device_id;             /* HIP device ordinal                         */
  double descriptor_min_len;     /* STDesc.cpp:181                             */
  double descriptor_max_len;     /* STDesc.cpp:180; *1000 must be < 2^21       */
  double std_side_resolution;    /* STDesc.cpp:178                             */
  double rough_dis_threshold;    /* STDesc.cpp:357                             */
  uint32_t first_frame_id;       /* initial current_frame_id_ (STDesc.h:363 is 0);
                                    a table shard of a multi-GPU map starts at
                                    the first frame it owns                    */
  uint32_t reserved;
} sgtd_config;

/* One descriptor = one STDesc (STDesc.h:75-97) without the unused covariance
 * matrices, as a structure of caller-allocated arrays.  NULL members are
 * skipped on output and read as zero on input (side, label, frame are
 * mandatory on input). */
typedef struct sgtd_desc_soa {
  double *side;      /* [n*3] side_length_ (scaled, ascending)                */
  double *angle;     /* [n*3] angle_                                          */
  double *center;    /* [n*3] center_                                         */
  float *vertex;     /* [n*9] vertex_A_,B_,C_ xyz (exact: they are f32 casts) */
  int32_t *label;    /* [n*3] (int)vertex_attached_                           */
  uint32_t *frame;   /* [n]   frame_id_                                       */
  int32_t *node_id;  /* [n*3] node_id = {i, m, n}                             */
} sgtd_desc_soa;

typedef struct sgtd_stats {
  int64_t n_entries;       /* E: descriptors in the table                      */
  int64_t n_buckets;       /* U: distinct (cell,label code) keys               */
  int64_t n_frames;        /* AddSTDescs calls so far                          */
  int64_t last_queries;    /* query frames in the last batch                   */
  int64_t last_D;          /* query descriptors in the last batch              */
  int64_t last_P;          /* table entries visited (STDesc.cpp:372 iterations)*/
  int64_t last_M;          /* rough matches (STDesc.cpp:378)                   */
  int64_t last_cand_pairs; /* pairs in all candidate match lists               */
  int64_t hbm_bytes_table; /* bytes of the hot (probed) table array (16 B/entry) */
  /* per-kernel device time of the last batch, ms (only when timing is enabled
   * with sgtd_set_timing; measured with hipEvents on the handle's stream)     */
  float ms_build, ms_sort, ms_probe, ms_votes, ms_topk, ms_count, ms_scan, ms_write, ms_total;
  int32_t overflowed;      /* last batch outgrew a work buffer and was re-run  */
  int32_t select_form;     /* how the last batch's match records were turned into votes and lists
                              (STDesc.cpp:404-453): 0 = one wave per 128-descriptor block (votes, topk,
                              block_count / _scan / _write), 1 = the lists by one workgroup per query
                              (pairs_query_kernel), 2 = votes + top-k by one workgroup per query as
                              well (votes_topk_kernel)                                        */
  int64_t last_P_swept;    /* table entries the sweep really loaded: last_P minus the
                              sub-cells of the visited buckets that no match can lie in */
  double bucket_len_sq_over_E; /* sum over buckets of len^2 / E: the bucket length a table entry
                              sees on average (sizes the first batch's work buffers)        */
  int64_t tail_entries;    /* entries in the tail segment (appended after the last full build
                              of the probe layout; 0 = one segment)                         */
  float ms_finalize;       /* wall time of the last probe-layout build: proportional to the
                              appended entries while they fit the tail segment              */
  float reserved2;
  /* since the handle was created (what a caller sees over a stream of DIFFERENT batches): */
  int64_t batches_total;     /* launches of a query batch's pipeline, counted on the device (re-runs included) —
                                the count includes batches that were enqueued and overwritten by the next one
                                without a sgtd_sync / sgtd_result_* call in between                   */
  int64_t overflow_launches_total; /* ... of them, launches that raised a work-buffer flag (their results are
                                incomplete until sgtd_sync has re-run them): a caller that enqueues batch after
                                batch without synchronising checks that this did not move            */
  int64_t reruns_total;      /* launches of a whole batch beyond the first (a work buffer
                                overflowed: match records, pass pool, GroupRows, undecided queue) */
  int64_t rewrites_total;    /* re-runs of the list pass alone (candidate-pair buffer too small) */
  int64_t list_moves_total;  /* match lists that outgrew the room they were given and moved to a
                                fresh slab during the sweep (probe_kernels.hip.h make_room)      */
  int64_t last_list_moves;   /* ... of the last batch                                         */
  int64_t device_allocs_total; /* hipMalloc calls of the handle's work and table buffers so far (a buffer that grows is
                                freed and allocated again, which synchronises the device: a timed region should see none) */
} sgtd_stats;

typedef struct sgtd_engine *sgtd_handle;

void sgtd_default_config(sgtd_config *cfg);
int sgtd_create(const sgtd_config *cfg, sgtd_handle *out);
/* One handle over several GPUs of this process: map frames are dealt to the
 * devices in blocks of 64 consecutive frames, round robin; every device holds a complete table
 * for its frames; a query batch is swept on all devices concurrently and the per-device
 * top-candidate_num tables are merged on the host with the reference's rule (votes descending,
 * ties -> lowest frame id, STDesc.cpp:423-433) — the single-table result.  Match lists, entries
 * and sgtd_verify stay with the owning device and are fetched from it; db_entry ids of such a
 * handle are opaque (owner in the upper bits) and valid for sgtd_fetch_entries only.
 * Host pointers only (device_ptrs must be 0); sgtd_add takes descriptors stamped with the
 * current frame id (what sgtd_build stamps).  Not available on it: sgtd_set_stream,
 * sgtd_export_*_dev, sgtd_result_rough, sgtd_table_dump, sgtd_save_table / sgtd_load_table
 * (SGTD_ERR_UNSUPPORTED).  n_dev == 1 gives an ordinary handle on device_ids[0];
 * cfg->device_id is ignored, cfg->first_frame_id must be a multiple of 64 * n_dev. */
int sgtd_create_multi(const sgtd_config *cfg, const int *device_ids, int n_dev, sgtd_handle *out);
/* devices behind the handle (1 for an ordinary handle) and the per-device engine k (borrowed;
 * e.g. for per-device sgtd_get_stats) */
int sgtd_device_count(sgtd_handle h);
sgtd_handle sgtd_device_handle(sgtd_handle h, int k);
int sgtd_destroy(sgtd_handle h);
const char *sgtd_strerror(int status);
/* text of the last HIP failure on this handle ("" if none) */
const char *sgtd_last_error(sgtd_handle h);

/* all work of the handle goes to this hipStream_t (NULL = the null stream) */
int sgtd_set_stream(sgtd_handle h, void *hip_stream);
/* enable per-kernel hipEvent timing (costs a few microseconds per launch) */
int sgtd_set_timing(sgtd_handle h, int enabled);

/* ---- key helpers: the pure functions the kernels derive their keys with (the same inline
 * code, instantiated on the host; no device needed) — pinned by the CPU tests against the
 * reference's own Combinatorial_Binary_Encoding (STDesc.cpp:3-16), STDesc_LOC equality
 * (STDesc.h:229-236: x, y, z, a only) and VOXEL_LOC equality (STDesc.h:133-135).
 * sgtd_table_key: cell coordinates below 65 536 and a 12-bit code; sgtd_dedup_key: millimetre
 * coordinates below 2^21 (the envelope check_cfg enforces).  Inside those envelopes two keys
 * are equal words exactly when the reference's operator== holds. */
uint32_t sgtd_label_code(int a, int b, int c);
uint64_t sgtd_table_key(uint32_t code, uint32_t x, uint32_t y, uint32_t z);
uint64_t sgtd_dedup_key(uint64_t mx, uint64_t my, uint64_t mz);

/* STDescManager::current_frame_id_ (STDesc.h:350) */
int sgtd_current_frame_id(sgtd_handle h, uint32_t *out);

/* ---- BuildSingleScanSTD (STDesc.cpp:174-315) --------------------------- */
/* One frame: keypoints xyz[n*3] f32 + label[n] (the pcl::PointXYZL fields,
 * utility.hpp:646-659), host memory.  Descriptors are stamped with the
 * current frame id.  out arrays must hold sgtd_max_descs(h, n) descriptors.
 * A frame with n < K yields 0 descriptors (the reference reads past its k-NN
 * result there). */
int64_t sgtd_max_descs(sgtd_handle h, int n_keypoints);
int sgtd_build(sgtd_handle h, const float *xyz, const uint32_t *label, int n,
               sgtd_desc_soa *out, int64_t capacity, int64_t *n_out);

/* ---- AddSTDescs (STDesc.cpp:149-172) ------------------------------------ */
/* Appends n host descriptors as one frame: current_frame_id_ is incremented
 * first, entries keep the frame id stored in d->frame. */
int sgtd_add(sgtd_handle h, const sgtd_desc_soa *d, int64_t n);

/* Map construction, the caller's loop semantic_graph_localization.cpp:419-458
 * for n_frames frames at once, entirely on the device: frame k (keypoints
 * kp_off[k]..kp_off[k+1]) is built with frame id current_frame_id_ and added,
 * then current_frame_id_++ .  xyz/label are host pointers unless
 * device_ptrs != 0; kp_off is always a host array of n_frames+1 offsets. */
int sgtd_add_frames(sgtd_handle h, const float *xyz, const uint32_t *label,
                    const int64_t *kp_off, int n_frames, int device_ptrs);

/* Sorts the appended entries into the probe layout (by key, sub-cells inside a bucket, bucket
 * directory + key hash).  Idempotent; called implicitly by the first query after an add.
 * Appending to a finalized table (AddSTDescs only ever appends, STDesc.cpp:149-172) sorts only
 * the appended entries, into a tail segment that every query sweeps after the main one; the
 * tail is merged into the main segment when it outgrows an eighth of it (sgtd_stats.tail_entries,
 * .ms_finalize). */
int sgtd_finalize(sgtd_handle h);

/* Two (or more) batches in flight over ONE map: `view` — a handle of the same configuration on the same device,
 * without a table of its own — borrows the finalized table of `owner` (cold entries, probe layout, entry ids) and keeps
 * its own work buffers, results and stream.  Batches enqueued alternately on the two handles' streams overlap on the
 * device: the descriptor build, home-cell sort and plan of one run beside the passes over the other's match records.
 * The owner must outlive its views (sgtd_destroy of an owner with views is SGTD_ERR_STATE); after anything that changes
 * the owner's table (sgtd_add*, sgtd_load_table, a finalize that merges a tail — the owner's own fifth batch on a tail
 * does) every call of a view that would touch the table again returns SGTD_ERR_STATE until it is attached again: the next
 * query, and for a batch still pending sgtd_sync / sgtd_verify* / sgtd_search_loop / the result calls that gather entries
 * (the view's batch is dropped; nothing reads the owner's freed buffers).  A view cannot add, load or rebuild (SGTD_ERR_STATE).  Neither handle
 * is thread safe; the table itself is only read by queries, so one host thread per handle is fine. */
int sgtd_attach_table(sgtd_handle view, sgtd_handle owner);

/* ---- candidate_selector (STDesc.cpp:318-460) ---------------------------- */
/* Fused query: for every query frame BuildSingleScanSTD (frame id =
 * current_frame_id_, as semantic_graph_localization.cpp:592) followed by
 * candidate_selector.  Enqueues on the stream and returns; results stay on the
 * device until fetched with the sgtd_result_* calls (which synchronise).
 * Inputs must stay valid until the first sgtd_result_* call returns. */
int sgtd_query_frames(sgtd_handle h, const float *xyz, const uint32_t *label,
                      const int64_t *kp_off, int n_queries, int device_ptrs);

/* candidate_selector on caller-provided descriptors (one query frame). */
int sgtd_query_descs(sgtd_handle h, const sgtd_desc_soa *q, int64_t nq);

/* Results of the last query batch.  For query q:
 *   n_cand[q]                      number of candidates (<= candidate_num)
 *   cand_frame/cand_votes[q*cn+k]  match_id_.second / vote count, in the
 *                                  reference's order (votes desc, frame asc)
 *   pair_off[q*(cn+1)+k]           offsets of candidate k's match_list_ into
 *                                  the pair arrays of query q
 * where cn = candidate_num.  Any pointer may be NULL. */
int sgtd_result_candidates(sgtd_handle h, int32_t *n_cand, int32_t *cand_frame,
                           int32_t *cand_votes, int64_t *pair_off);
/* Asynchronous device-to-device export of the candidate tables of the last
 * batch into caller device buffers (int32 [n_queries*candidate_num] each;
 * unused slots hold frame -1 / votes 0), enqueued on the handle's stream
 * without synchronising: the multi-GPU path all-gathers them with RCCL. */
int sgtd_export_candidates_dev(sgtd_handle h, int32_t *d_cand_frame, int32_t *d_cand_votes);
/* ---- the multi-GPU step (one process per GPU, sgtd_amd/dist.py): every rank holds the table of a
 * frame range (sgtd_config.first_frame_id), sweeps it with the query batch and takes its local top-candidate_num; the
 * local tables travel in ONE all-gather (RCCL) per table group and every rank merges them with the reference's rule
 * (the largest vote count first, ties -> the lowest frame id, >= 5 votes: STDesc.cpp:423-433) — the list a single table
 * gives.  The calls below keep that exchange off the critical path:
 *   sgtd_set_candidate_export   registers a device buffer of sgtd_candidate_export_ints(n_queries, candidate_num) int32;
 *                               from then on every batch writes its local tables there, packed [frames | votes | 4 flag
 *                               words], as soon as they are final — BEFORE its match lists are written (NULL: off)
 *   sgtd_export_wait            makes side_stream wait (on the device) for the last enqueued batch's packed table
 *   sgtd_export_release         recorded on side_stream behind the last reader of the packed table (the all-gather): the
 *                               next batch's export waits for it — neither call blocks the host
 *   sgtd_merge_candidates_dev   the merge as one kernel on `stream`: d_gathered = n_tables packed tables back to back (an
 *                               all-gather's output); outputs int32 [n_queries*candidate_num] frames / votes (unused: -1 / 0)
 *                               and src (table << 8 | slot in the owner's local list, -1 unused), n_cand [n_queries], keep
 *                               u64 [n_queries] (bit s: slot s of table my_table's local list survived; my_table -1: none)
 *                               and flags[0] (bit 0: some table came from a batch that outgrew a work buffer — sgtd_sync
 *                               every rank and exchange again; bit 1: tables of different shapes)
 *   sgtd_set_deferred_lists /   a batch stops behind its candidate tables; sgtd_finish_lists writes the match lists — of the
 *   sgtd_finish_lists           candidates in d_keep only (NULL: all).  Repeatable (the match records stay intact).  Batches
 *                               whose lists come from the per-block passes (small batches, sgtd_stats.select_form 0) ignore
 *                               both: they always hold every local candidate's list.
 *   sgtd_verify_masked          sgtd_verify for the candidates in d_keep only (the others score -1)
 *   sgtd_gather_verified_dev    verification results of the merged candidates out of the owners' tables: d_gathered =
 *                               n_tables x [score f64 n_queries*cn | pose f64 n_queries*cn*12] (every rank's
 *                               sgtd_export_verify_dev output, all-gathered), d_src = the merge's src
 * Not available on a multi-device handle (that one merges on the host). */
int64_t sgtd_candidate_export_ints(int n_queries, int cand_num);
int sgtd_set_candidate_export(sgtd_handle h, int32_t *d_packed, int64_t capacity_ints);
int sgtd_export_wait(sgtd_handle h, void *side_stream);
int sgtd_export_release(sgtd_handle h, void *side_stream);
int sgtd_merge_candidates_dev(sgtd_handle h, void *stream, const int32_t *d_gathered, int n_tables, int my_table, int n_queries,
                              int32_t *d_frame, int32_t *d_votes, int32_t *d_n_cand, int32_t *d_src, uint64_t *d_keep,
                              int32_t *d_flags);
int sgtd_gather_verified_dev(sgtd_handle h, void *stream, const double *d_gathered, int n_tables, const int32_t *d_src,
                             int n_queries, double *d_score, double *d_pose);
int sgtd_set_deferred_lists(sgtd_handle h, int on);
int sgtd_finish_lists(sgtd_handle h, const uint64_t *d_keep);
int sgtd_verify_masked(sgtd_handle h, const uint64_t *d_keep);

/* number of query descriptors of query q (stds_vec.size()) */
int sgtd_result_query_desc_count(sgtd_handle h, int q, int64_t *n);
/* match_list_ pairs of query q, candidate after candidate, each in the
 * reference's order: q_idx = index into the query's descriptors, db_entry =
 * insertion index of the table entry (use sgtd_fetch_entries).  capacity in
 * pairs; *n_pairs receives the total. */
int sgtd_result_pairs(sgtd_handle h, int q, int32_t *q_idx, int64_t *db_entry,
                      int64_t capacity, int64_t *n_pairs);
/* the query's own descriptors (built on the device by sgtd_query_frames) */
int sgtd_result_query_descs(sgtd_handle h, int q, sgtd_desc_soa *out,
                            int64_t capacity, int64_t *n_out);
/* per-frame vote counts of query q (match_array, STDesc.cpp:323,410):
 * votes[f - frame_lo] for f in [frame_lo, frame_lo + n) */
int sgtd_result_votes(sgtd_handle h, int q, uint32_t *votes, int64_t capacity,
                      uint32_t *frame_lo, int64_t *n);
/* all rough matches of query q in the reference's (i, cell, j) order
 * (STDesc.cpp:378-384): q_idx, voxel_round index 0..26, db_entry, frame, dis.
 * Diagnostic / parity output; any pointer may be NULL.  The first call after a batch re-runs
 * the batch with the diagnostic sweep (the reference's distance test verbatim on the exact f64
 * sides, cell index and distance recorded per match); the product sweep keeps neither. */
int sgtd_result_rough(sgtd_handle h, int q, int32_t *q_idx, int32_t *cell,
                      int64_t *db_entry, uint32_t *frame, double *dis,
                      int64_t capacity, int64_t *n_rough);

/* ---- geometric verification of the candidates of the last batch ----
 * STDescManager::candidate_verify + triangle_solver (STDesc.cpp:462-571) for every
 * (query, candidate) of the batch, on the device: hypotheses = every skip_len-th pair of the
 * candidate's match_list_ (:467-468), votes with the 3.0 inlier radius (:469-505), first
 * maximum, >= 4 votes (:507-515), inliers of the best hypothesis (:516-539).
 * The 3x3 solver is a one-sided Jacobi SVD, not Eigen::JacobiSVD (not available here):
 * parity with the reference binary's SVD is unpinned (rotations agree to rounding, inlier
 * sets can differ only for pairs within rounding of the 3.0 radius). */
int sgtd_verify(sgtd_handle h);
/* verify_score (inlier count, -1 = rejected, :539-541) and relative_pose of every candidate
 * of query q: score[candidate_num], pose[candidate_num*12] = rot row-major (9) then t (3);
 * entries past the query's candidate count hold -1 / zeros.  Either pointer may be NULL. */
int sgtd_result_verify(sgtd_handle h, int q, double *score, double *pose);
/* asynchronous device-to-device export of the verification results of the whole batch into
 * caller device buffers (score f64 [n_queries*candidate_num], pose f64 [n_queries*candidate_num*12]),
 * enqueued on the handle's stream without synchronising: the table-sharded multi-GPU path
 * all-gathers them with RCCL (every candidate frame is verified by the rank that owns it) */
int sgtd_export_verify_dev(sgtd_handle h, double *d_score, double *d_pose);
/* sucess_match_vec of (query q, candidate cand) as positions into that candidate's
 * match_list_ (ascending = list order); capacity in elements, *n = needed */
int sgtd_result_inliers(sgtd_handle h, int q, int cand, int32_t *idx, int64_t capacity, int64_t *n);
/* The inlier pairs of EVERY candidate of query q in one call (one device compaction, one copy):
 * candidate k's pairs are [cand_off[k], cand_off[k+1]) of q_idx / db_entry, in match-list order;
 * cand_off holds candidate_num + 1 offsets.  What SearchLoop needs to fill
 * LOOP_RESULT::loop_std_pair (STDesc.cpp:119-124) without fetching the full match lists. */
int sgtd_result_inlier_pairs(sgtd_handle h, int q, int64_t *cand_off, int32_t *q_idx, int64_t *db_entry,
                             int64_t capacity, int64_t *n_pairs);
/* The same pairs with the table side already fetched: q_idx[i] and entry i of `entries` (the fields whose
 * pointers are set) are the i-th inlier pair — the compaction, the gather of the table entries and the copies
 * run back to back on the device, no list of indices travels to the host and back.  capacity in pairs
 * (the sum of the candidates' list lengths, sgtd_result_candidates' last offset, always suffices);
 * *n_pairs = needed.  SGTD_ERR_CAPACITY leaves cand_off and *n_pairs valid. */
int sgtd_result_inlier_entries(sgtd_handle h, int q, int64_t *cand_off, int32_t *q_idx, sgtd_desc_soa *entries,
                               int64_t capacity, int64_t *n_pairs);
/* STDescManager::SearchLoop's choice (STDesc.cpp:105-146) for every query of the batch:
 * the first candidate with the strictly largest verify_score, accepted if it exceeds
 * icp_threshold; best_frame = -1 and best_score = 0 otherwise (loop_result (-1, 0)).
 * Arrays of n_queries; any may be NULL.  Requires sgtd_verify. */
int sgtd_search_loop(sgtd_handle h, double icp_threshold, int32_t *best_cand, int32_t *best_frame,
                     double *best_score);

/* One query frame through candidate_selector (STDesc.cpp:318-460), candidate_verify for every candidate (:462-571) and the
 * inlier pairs of every candidate with the table entries they name — what SearchLoop (:84-147) needs — in ONE call: the
 * reference's one-frame-per-call pattern (semantic_graph_localization.cpp:590-603).  Equal, value for value, to
 * sgtd_query_descs + sgtd_verify + sgtd_result_candidates + sgtd_result_verify(0) + sgtd_result_inlier_entries(0), which
 * wait for the device eight times and issue some sixty small copies between them; this call enqueues everything behind
 * the batch and waits once: the small results arrive as one packed block, the entries where the caller wants them.  Output arrays of
 * candidate_num (pair_off, inlier_off: candidate_num + 1; pose: candidate_num * 12) elements; any pointer may be NULL.
 * capacity = room (pairs) in inlier_q_idx / entries; n_inliers = needed.  SGTD_ERR_CAPACITY leaves everything but the
 * inlier pairs valid — sgtd_result_inlier_entries(h, 0, ...) with more room fetches them.  The handle afterwards is in
 * the state the five calls leave (every sgtd_result_* call works).  Not on a multi-device handle (SGTD_ERR_UNSUPPORTED).
 * inlier_q_idx and the members of `entries` are best given as sgtd_host_alloc memory (page-locked, `capacity` entries each): the
 * device then writes the inlier pairs in place and the call has one wait; ordinary arrays are filled from a page-locked block of
 * the handle's with memcpy (same results).  On SGTD_ERR_CAPACITY the first `capacity` pairs may or may not have been written.
 * flags = SGTD_FRAME_LISTS_ONLY: candidate_selector ALONE (STDesc.cpp:318-460), for a node that keeps its own candidate_verify —
 * no verification (score and pose are not written; the handle is left as sgtd_query_descs leaves it), and the pairs handed
 * back are ALL pairs of every candidate's match_list_, in the reference's order: inlier_off = pair_off, n_inliers = their
 * number, inlier_q_idx / entries = the query descriptor and the table entry of every pair.  Equal to sgtd_query_descs +
 * sgtd_result_candidates + sgtd_result_pairs + sgtd_fetch_entries, in one call and one wait.  On SGTD_ERR_CAPACITY call again
 * with room for n_inliers pairs. */
#define SGTD_FRAME_LISTS_ONLY 1   /* sgtd_frame_search.flags: candidate_selector alone (below) */
typedef struct sgtd_frame_search {
  int32_t n_cand;           /* out */
  int32_t flags;            /* in: 0, or SGTD_FRAME_LISTS_ONLY (the field was `reserved`, always 0, before)                 */
  int32_t *cand_frame;      /* out [candidate_num]: match_id_.second, votes descending / frame ascending            */
  int32_t *cand_votes;      /* out [candidate_num]                                                                  */
  int64_t *pair_off;        /* out [candidate_num + 1]: offsets of the candidates' match lists (their lengths)      */
  double *score;            /* out [candidate_num]: verify_score (-1: rejected)                                     */
  double *pose;             /* out [candidate_num * 12]: rot row-major (9), t (3)                                   */
  int64_t *inlier_off;      /* out [candidate_num + 1]: candidate k's inlier pairs are [inlier_off[k], inlier_off[k + 1]) */
  int32_t *inlier_q_idx;    /* out [capacity]: query descriptor of inlier pair i                                    */
  sgtd_desc_soa entries;    /* out: table entry of inlier pair i (arrays of capacity descriptors; NULL members skipped) */
  int64_t capacity;         /* in                                                                                   */
  int64_t n_inliers;        /* out                                                                                  */
} sgtd_frame_search;
int sgtd_search_frame(sgtd_handle h, const sgtd_desc_soa *q, int64_t nq, sgtd_frame_search *io);

/* ---- persistent table ----
 * The reference rebuilds data_base_ from the map files at every start
 * (semantic_graph_localization.cpp:419-458) and only ever appends (STDesc.cpp:149-172).
 * sgtd_save_table writes the table in insertion order (all descriptor fields, the frame
 * counter, the frame range); sgtd_load_table replaces the handle's table with a saved one —
 * the probe layout is re-derived by the next query (a sort of ~1 ms per 4 M entries), and
 * sgtd_add / sgtd_add_frames keep appending after it (new session on an old map).  The file
 * records std_side_resolution (sides are stored scaled): loading into a handle configured
 * differently is SGTD_ERR_INVALID.  I/O problems are SGTD_ERR_IO (sgtd_last_error names the file). */
int sgtd_save_table(sgtd_handle h, const char *path);
int sgtd_load_table(sgtd_handle h, const char *path);

/* ---- graph-JSON ingest (host code, no device needed) ----
 * readGraphFromFile / fromJSON (include/Semantic_Graph.hpp:122-184) + Graph2CloudL
 * (include/utility.hpp:646-659) for many files at once: {"nodes":[int], "centers":[[x,y,z]],
 * "poses":[12 floats], ...} (producer get_json.cpp:332-341; only these three keys are read;
 * numbers as nlohmann::json stores and casts them — integer tokens through strtoull / strtoll, the rest through
 * strtod, then get<float>() / get<int>() — pinned bit for bit against that library by
 * tests/cpp/test_ingest_nlohmann.cpp; a key that occurs twice keeps its LAST value like nlohmann::json 3.2 and later, its
 * first like 3.1.1 when SGTD_JSON_DUPLICATE_KEYS=first is set in the environment) parsed on
 * n_threads host threads into the arrays sgtd_add_frames / sgtd_query_frames take.  Frames
 * keep the order of `paths`.  On SGTD_ERR_IO *out still holds an object whose
 * sgtd_graphs_error() names the file ("Error opening file: ..." like Semantic_Graph.hpp:173);
 * free it with sgtd_graphs_free. */
typedef struct sgtd_graph_batch sgtd_graph_batch;
int sgtd_graphs_load(const char *const *paths, int n_files, int n_threads, sgtd_graph_batch **out);
/* binary cache of a parsed batch (skips JSON parsing at the next start) */
int sgtd_graphs_save_cache(const sgtd_graph_batch *b, const char *path);
int sgtd_graphs_load_cache(const char *path, sgtd_graph_batch **out);
/* borrowed pointers, valid until sgtd_graphs_free: xyz f32 [n_keypoints*3], label u32
 * [n_keypoints], kp_off i64 [n_frames+1], poses f32 [n_frames*12]; any pointer may be NULL */
int sgtd_graphs_view(const sgtd_graph_batch *b, int *n_frames, int64_t *n_keypoints, const float **xyz,
                     const uint32_t **label, const int64_t **kp_off, const float **poses);
const char *sgtd_graphs_error(const sgtd_graph_batch *b);
void sgtd_graphs_free(sgtd_graph_batch *b);

/* Page-locked host memory for the destination buffers of the fetch calls (sgtd_fetch_entries,
 * sgtd_result_pairs, ...): a copy into it is one direct DMA transfer — pageable memory goes through the
 * runtime's staging at about a third of the rate.  Allocation is slow (milliseconds): keep the buffers
 * across calls, as adapter/STDesc_shim.hpp does.  bytes == 0 yields *out = NULL. */
int sgtd_host_alloc(size_t bytes, void **out);
int sgtd_host_free(void *p);

/* table entries by insertion index (to rebuild pair<STDesc,STDesc>) */
int sgtd_fetch_entries(sgtd_handle h, const int64_t *db_entry, int64_t n,
                       sgtd_desc_soa *out);

/* table layout for inspection: keys [U*4] = x,y,z,label code ascending by
 * (code,x,y,z); bucket_off [U+1]; entry ids [E] in bucket order */
int sgtd_table_dump(sgtd_handle h, int64_t *keys, int64_t *bucket_off,
                    int64_t *entry_ids, int64_t cap_buckets, int64_t cap_entries);

/* Largest number of query frames of n_keypoints keypoints each that one sgtd_query_frames call
 * should carry: the rough matches of a batch are indexed with 32 bits (more returns
 * SGTD_ERR_CAPACITY) and their records must fit the device memory that is free now.  Uses the
 * matches per query of the last batch when there is one, else an estimate from the table's
 * bucket statistics; conservative by a factor of about two. */
int sgtd_max_batch(sgtd_handle h, int n_keypoints, int64_t *max_queries);

/* waits for the stream, re-runs the last batch with larger work buffers if it
 * overflowed them, and fills the counters */
int sgtd_sync(sgtd_handle h);
int sgtd_get_stats(sgtd_handle h, sgtd_stats *out);

#ifdef __cplusplus
}
#endif
#endif
