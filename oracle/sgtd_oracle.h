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

/* ---- parity-risk audit (tools/parity_audit.py): how many of the path's DECISIONS would come out differently under
 * arithmetic the real binary might use where this restatement had to infer it — Eigen's
 * Vector3d::norm() association, FMA contraction, +-1 ulp on dis_threshold, FLANN's order among exactly tied
 * neighbours — and how close the closest call is.  Counters accumulate over calls (zero the struct first).
 * Variants of a 3-term sum of squares: [0] right association  sqrt(v0^2 + (v1^2 + v2^2))
 *                                       [1] left association with both adds contracted  fma(v2, v2, fma(v1, v1, v0*v0))
 *                                       [2] right association contracted                fma(v0, v0, fma(v1, v1, v2*v2))
 * (the reference form is the left association with every operation rounded: sqrt((v0^2 + v1^2) + v2^2)). */
typedef struct orc_audit {
  /* candidate_selector's loop (STDesc.cpp:351-400) */
  int64_t gate_tests;            /* (query descriptor, cell) tests of :366-369                                   */
  int64_t gate_flips[3];         /* ... whose outcome differs under variant k                                    */
  int64_t gate_flip_visits[3];   /* table entries in the buckets of those cells                                  */
  int64_t visits;                /* (query descriptor, table entry) distance tests (:374-378, other frame)       */
  int64_t match_flips[3];        /* ... whose outcome differs with BOTH norms (dis and dis_threshold) of variant k */
  int64_t thr_ulp_flips[2];      /* ... with the reference's dis against dis_threshold + 1 ulp / - 1 ulp          */
  int64_t near_calls;            /* ... with |dis - dis_threshold| <= 64 ulp(dis_threshold)                       */
  double min_margin;             /* min |dis - dis_threshold| over all visits                                    */
  double min_margin_ulps;        /* ... in ulps of dis_threshold                                                 */
  double min_gate_margin;        /* min | ||side - centre|| - 1.5 | over all gate tests                          */
  /* BuildSingleScanSTD (STDesc.cpp:183-308) */
  int64_t knn_points;            /* keypoints                                                                    */
  int64_t knn_tied_points;       /* ... with an exact f32 tie among the squared distances of their K + 1 nearest  */
  int64_t knn_fma_order_diffs;   /* ... whose ordered K-NN list differs when FLANN's accumulation is contracted   */
  int64_t triplets;              /* enumerated (i, m, n)                                                         */
  int64_t side_value_diffs[3];   /* sides (3 per triplet) whose f64 value differs under variant k                */
  int64_t build_flips[3];        /* triplets where a length limit (:204-208), a sort comparison (:220-243), a millimetre
                                    key (:246-250), an insert cell (:155-157) or a probe cell (:359-361) differs     */
  double min_len_margin;         /* closest side to descriptor_min_len / descriptor_max_len                       */
  double min_cell_margin;        /* closest scaled side to a cell boundary (k for the probe, k + 0.5 for the insert) */
} orc_audit;
/* audits BuildSingleScanSTD on one frame (and builds it: the frame becomes "last built") */
void orc_audit_build(orc_manager *m, const float *xyz, const uint32_t *label, int n, orc_audit *acc);
/* audits candidate_selector's loop for the last built frame against the table (results are not kept) */
void orc_audit_select(orc_manager *m, orc_audit *acc);

/* ---- the same for candidate_verify (STDesc.cpp:462-547): Eigen::JacobiSVD is the third inferred piece (this restatement
 * solves the 3x3 problem with a one-sided Jacobi SVD).  The caller hands in the hypotheses (R row-major, t: 12 doubles each)
 * another SVD gives for the same covariance matrices — tools/parity_audit.py --verify uses LAPACK's through numpy — and
 * every decision of the candidate's verification is evaluated with both. */
typedef struct orc_verify_audit {
  int64_t candidates, hypotheses, pair_tests;   /* pair_tests = (pair, hypothesis) combinations (:488-505)              */
  int64_t vertex_tests;                         /* vertex distance tests evaluated (all three of every combination)     */
  int64_t vertex_flips, pair_flips;             /* ... whose outcome differs with the other SVD's (R, t)                */
  int64_t vote_list_diffs;                      /* hypotheses whose vote count differs                                  */
  int64_t best_index_diffs, score_diffs, inlier_set_diffs;   /* candidates whose :507-514 choice / :539 score / :516-539 set differs */
  int64_t near_calls;                           /* vertex tests with | ||d|| - 3 | <= 1e-9                              */
  double min_margin;                            /* min | ||d|| - 3 | over all vertex tests                             */
  double max_norm_diff;                         /* max | ||d||_this - ||d||_other |                                    */
  double max_rot_diff, max_t_diff;              /* largest entry difference between the two solutions                  */
} orc_verify_audit;
/* use_size of candidate `cand` of the last select (0: none) and, if cov != NULL, per hypothesis the covariance matrix
 * (:558, row-major), the query centre and the table centre */
int orc_verify_hyp_inputs(orc_manager *m, int cand, double *cov, double *qc, double *ec);
/* the restatement's own hypotheses of the candidate, [use_size][12] (R row-major, t) */
void orc_verify_hyp_solutions(orc_manager *m, int cand, double *rt);
void orc_audit_verify(orc_manager *m, int cand, const double *other_rt, int n_hyp, orc_verify_audit *acc);

#ifdef __cplusplus
}
#endif
#endif
