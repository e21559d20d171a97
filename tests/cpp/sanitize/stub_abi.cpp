// stub_abi.cpp — a HOST-ONLY stand-in for the part of the C ABI (include/sgtd_accel.h) that adapter/STDesc_shim.hpp
// calls, so that the adapter's own code — buffer sizing, the page-locked entry buffers, the fill team of short-lived
// threads, the in-place construction of ~10^5 pair<STDesc, STDesc> per call — runs under AddressSanitizer /
// UndefinedBehaviorSanitizer / ThreadSanitizer on the CPU (no GPU sanitizer exists on this pool).  Test infrastructure:
// nothing here computes a descriptor or a match; the "results" are deterministic pseudo-random tables of the right
// SHAPE (candidate counts, list lengths of 10^3..10^4 pairs, inlier subsets), every caller buffer is written exactly as
// far as the real library writes it, capacities are honoured the same way, and every index handed out is in range —
// so any out-of-bounds access the sanitizers report is the adapter's.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../../include/sgtd_accel.h"

struct sgtd_engine {
  sgtd_config cfg;
  uint32_t frame = 0;
  // the "table": every descriptor handed to sgtd_add, structure of arrays
  std::vector<double> side, angle, center;
  std::vector<float> vertex;
  std::vector<int32_t> label, node_id;
  std::vector<uint32_t> fr;
  // last query
  int64_t nq = 0;
  int n_cand = 0;
  std::vector<int32_t> cand_frame, cand_votes, q_idx;
  std::vector<int64_t> off, entry;
  std::vector<double> score;
  bool verified = false;
  uint64_t rng = 88172645463325252ull;
  uint64_t next() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; }
  int64_t n_entries() const { return (int64_t)fr.size(); }
};

static void copy_entry(const sgtd_engine *e, int64_t g, sgtd_desc_soa *out, int64_t i) {
  if (out->side) memcpy(out->side + 3 * i, e->side.data() + 3 * g, 3 * sizeof(double));
  if (out->angle) memcpy(out->angle + 3 * i, e->angle.data() + 3 * g, 3 * sizeof(double));
  if (out->center) memcpy(out->center + 3 * i, e->center.data() + 3 * g, 3 * sizeof(double));
  if (out->vertex) memcpy(out->vertex + 9 * i, e->vertex.data() + 9 * g, 9 * sizeof(float));
  if (out->label) memcpy(out->label + 3 * i, e->label.data() + 3 * g, 3 * sizeof(int32_t));
  if (out->frame) out->frame[i] = e->fr[(size_t)g];
  if (out->node_id) memcpy(out->node_id + 3 * i, e->node_id.data() + 3 * g, 3 * sizeof(int32_t));
}

extern "C" {

void sgtd_default_config(sgtd_config *c) {
  memset(c, 0, sizeof(*c));
  c->descriptor_near_num = 10; c->candidate_num = 50; c->max_frame_n = 20000;
  c->descriptor_min_len = 0.5; c->descriptor_max_len = 50.0; c->std_side_resolution = 1.0; c->rough_dis_threshold = 0.03;
}
int sgtd_create(const sgtd_config *cfg, sgtd_handle *out) {
  if (!cfg || !out) return SGTD_ERR_INVALID;
  sgtd_engine *e = new sgtd_engine();
  e->cfg = *cfg;
  *out = e;
  return SGTD_OK;
}
int sgtd_create_multi(const sgtd_config *cfg, const int *ids, int n, sgtd_handle *out) {
  if (!ids || n < 1) return SGTD_ERR_INVALID;
  return sgtd_create(cfg, out);
}
int sgtd_destroy(sgtd_handle h) { delete h; return SGTD_OK; }
const char *sgtd_strerror(int) { return "stub"; }
int sgtd_current_frame_id(sgtd_handle h, uint32_t *out) { *out = h->frame; return SGTD_OK; }
int64_t sgtd_max_descs(sgtd_handle h, int n) {
  const int k = h->cfg.descriptor_near_num;
  return (int64_t)n * ((k - 1) * (k - 2) / 2);
}

// n_out = about 60 % of the bound, every written field derived from the keypoints
int sgtd_build(sgtd_handle h, const float *xyz, const uint32_t *label, int n, sgtd_desc_soa *out, int64_t capacity, int64_t *n_out) {
  if (!h || !out || !n_out || n < 0) return SGTD_ERR_INVALID;
  const int64_t want = sgtd_max_descs(h, n) * 6 / 10;
  *n_out = want;
  if (want > capacity) return SGTD_ERR_CAPACITY;
  for (int64_t i = 0; i < want; i++) {
    const int a = (int)(i % n), b = (int)((i * 7 + 1) % n), c = (int)((i * 13 + 2) % n);
    const int v[3] = {a, b, c};
    for (int k = 0; k < 3; k++) {
      out->side[3 * i + k] = 1.0 + (double)((i * 31 + k * 17) % 4000) / 100.0;
      if (out->angle) out->angle[3 * i + k] = 0.25 * (k + 1);
      if (out->center) out->center[3 * i + k] = (xyz[3 * a + k] + xyz[3 * b + k] + xyz[3 * c + k]) / 3.0;
      if (out->label) out->label[3 * i + k] = (int32_t)label[v[k]];
      if (out->node_id) out->node_id[3 * i + k] = v[k];
      for (int d = 0; d < 3; d++)
        if (out->vertex) out->vertex[9 * i + 3 * k + d] = xyz[3 * v[k] + d];
    }
    if (out->frame) out->frame[i] = h->frame;
  }
  return SGTD_OK;
}

int sgtd_add(sgtd_handle h, const sgtd_desc_soa *d, int64_t n) {
  if (!h || n < 0 || (n > 0 && (!d || !d->side || !d->label || !d->frame))) return SGTD_ERR_INVALID;
  h->frame++;
  for (int64_t i = 0; i < n; i++) {
    for (int k = 0; k < 3; k++) {
      h->side.push_back(d->side[3 * i + k]);
      h->angle.push_back(d->angle ? d->angle[3 * i + k] : 0.0);
      h->center.push_back(d->center ? d->center[3 * i + k] : 0.0);
      h->label.push_back(d->label[3 * i + k]);
      h->node_id.push_back(d->node_id ? d->node_id[3 * i + k] : 0);
    }
    for (int k = 0; k < 9; k++) h->vertex.push_back(d->vertex ? d->vertex[9 * i + k] : 0.f);
    h->fr.push_back(d->frame[i]);
  }
  return SGTD_OK;
}

int sgtd_query_descs(sgtd_handle h, const sgtd_desc_soa *q, int64_t nq) {
  if (!h || nq < 0 || (nq > 0 && (!q || !q->side || !q->label || !q->frame))) return SGTD_ERR_INVALID;
  const int cn = h->cfg.candidate_num;
  h->nq = nq;
  h->verified = false;
  const int64_t E = h->n_entries();
  // 0 .. cn candidates (sometimes none, sometimes all), lists of 0 .. ~6000 pairs (a few empty ones)
  h->n_cand = (nq == 0 || E == 0) ? 0 : (int)(h->next() % 8 == 0 ? 0 : 1 + h->next() % (uint64_t)cn);
  h->cand_frame.assign(cn, -1); h->cand_votes.assign(cn, 0); h->off.assign(cn + 1, 0);
  h->q_idx.clear(); h->entry.clear();
  for (int k = 0; k < h->n_cand; k++) {
    const int64_t len = h->next() % 16 == 0 ? 5 : 5 + (int64_t)(h->next() % 6000);
    h->cand_frame[k] = (int32_t)(h->next() % std::max<uint32_t>(h->frame, 1));
    h->cand_votes[k] = (int32_t)len;
    for (int64_t r = 0; r < len; r++) {
      h->q_idx.push_back((int32_t)(h->next() % (uint64_t)nq));
      h->entry.push_back((int64_t)(h->next() % (uint64_t)E));
    }
    h->off[k + 1] = h->off[k] + len;
  }
  for (int k = h->n_cand; k < cn; k++) h->off[k + 1] = h->off[h->n_cand];
  return SGTD_OK;
}

int sgtd_result_candidates(sgtd_handle h, int32_t *n_cand, int32_t *cand_frame, int32_t *cand_votes, int64_t *pair_off) {
  const int cn = h->cfg.candidate_num;
  if (n_cand) *n_cand = h->n_cand;
  if (cand_frame) memcpy(cand_frame, h->cand_frame.data(), cn * sizeof(int32_t));
  if (cand_votes) memcpy(cand_votes, h->cand_votes.data(), cn * sizeof(int32_t));
  if (pair_off) memcpy(pair_off, h->off.data(), (cn + 1) * sizeof(int64_t));
  return SGTD_OK;
}

int sgtd_result_pairs(sgtd_handle h, int q, int32_t *q_idx, int64_t *db_entry, int64_t capacity, int64_t *n_pairs) {
  if (q != 0 || !n_pairs) return SGTD_ERR_INVALID;
  const int64_t n = (int64_t)h->entry.size();
  *n_pairs = n;
  if (n > capacity) return SGTD_ERR_CAPACITY;
  if (q_idx && n) memcpy(q_idx, h->q_idx.data(), n * sizeof(int32_t));
  if (db_entry && n) memcpy(db_entry, h->entry.data(), n * sizeof(int64_t));
  return SGTD_OK;
}

int sgtd_fetch_entries(sgtd_handle h, const int64_t *db_entry, int64_t n, sgtd_desc_soa *out) {
  if (n < 0 || (n > 0 && (!db_entry || !out))) return SGTD_ERR_INVALID;
  for (int64_t i = 0; i < n; i++) {
    if (db_entry[i] < 0 || db_entry[i] >= h->n_entries()) return SGTD_ERR_INVALID;
    copy_entry(h, db_entry[i], out, i);
  }
  return SGTD_OK;
}

int sgtd_verify(sgtd_handle h) {
  const int cn = h->cfg.candidate_num;
  h->score.assign(cn, -1.0);
  for (int k = 0; k < h->n_cand; k++) {
    const int64_t len = h->off[k + 1] - h->off[k];
    h->score[k] = (h->next() % 5 == 0 || len < 8) ? -1.0 : (double)((len + 2) / 3);   // = the inliers handed out below
  }
  h->verified = true;
  return SGTD_OK;
}

int sgtd_result_verify(sgtd_handle h, int q, double *score, double *pose) {
  if (!h->verified || q != 0) return SGTD_ERR_INVALID;
  const int cn = h->cfg.candidate_num;
  if (score) memcpy(score, h->score.data(), cn * sizeof(double));
  if (pose)
    for (int i = 0; i < cn * 12; i++) pose[i] = (i % 12 == 0 || i % 12 == 4 || i % 12 == 8) ? 1.0 : 0.01 * (i % 12);
  return SGTD_OK;
}

// inliers of a verified candidate: every third pair of its list
int sgtd_result_inlier_entries(sgtd_handle h, int q, int64_t *cand_off, int32_t *q_idx, sgtd_desc_soa *entries, int64_t capacity,
                               int64_t *n_pairs) {
  if (!h->verified || q != 0 || !cand_off || !n_pairs) return SGTD_ERR_INVALID;
  const int cn = h->cfg.candidate_num;
  std::vector<int64_t> pick;
  cand_off[0] = 0;
  for (int k = 0; k < cn; k++) {
    if (k < h->n_cand && h->score[k] >= 0)
      for (int64_t r = h->off[k]; r < h->off[k + 1]; r += 3) pick.push_back(r);
    cand_off[k + 1] = (int64_t)pick.size();
  }
  *n_pairs = (int64_t)pick.size();
  if (*n_pairs > capacity) return SGTD_ERR_CAPACITY;
  for (size_t i = 0; i < pick.size(); i++) {
    if (q_idx) q_idx[i] = h->q_idx[(size_t)pick[i]];
    if (entries) copy_entry(h, h->entry[(size_t)pick[i]], entries, (int64_t)i);
  }
  return SGTD_OK;
}

int sgtd_search_frame(sgtd_handle h, const sgtd_desc_soa *q, int64_t nq, sgtd_frame_search *io) {
  if (!h || !io) return SGTD_ERR_INVALID;
  int st = sgtd_query_descs(h, q, nq);
  if (st != SGTD_OK) return st;
  if (io->flags & SGTD_FRAME_LISTS_ONLY) {         // candidate_selector alone: every pair of every list, no verification
    std::vector<int64_t> off((size_t)h->cfg.candidate_num + 1);
    if ((st = sgtd_result_candidates(h, &io->n_cand, io->cand_frame, io->cand_votes, off.data())) != SGTD_OK) return st;
    if (io->pair_off) memcpy(io->pair_off, off.data(), off.size() * sizeof(int64_t));
    if (io->inlier_off) memcpy(io->inlier_off, off.data(), off.size() * sizeof(int64_t));
    const int64_t n = (int64_t)h->entry.size();
    io->n_inliers = n;
    if (n > io->capacity) return SGTD_ERR_CAPACITY;
    for (int64_t i = 0; i < n; i++) {
      if (io->inlier_q_idx) io->inlier_q_idx[i] = h->q_idx[(size_t)i];
      copy_entry(h, h->entry[(size_t)i], &io->entries, i);
    }
    return SGTD_OK;
  }
  if ((st = sgtd_verify(h)) != SGTD_OK) return st;
  if ((st = sgtd_result_candidates(h, &io->n_cand, io->cand_frame, io->cand_votes, io->pair_off)) != SGTD_OK) return st;
  if ((st = sgtd_result_verify(h, 0, io->score, io->pose)) != SGTD_OK) return st;
  std::vector<int64_t> off((size_t)h->cfg.candidate_num + 1);
  st = sgtd_result_inlier_entries(h, 0, off.data(), io->inlier_q_idx, &io->entries, io->capacity, &io->n_inliers);
  if (io->inlier_off) memcpy(io->inlier_off, off.data(), off.size() * sizeof(int64_t));
  return st;
}

// "page-locked" memory: plain heap blocks, so that the sanitizers see every byte of them
int sgtd_host_alloc(size_t bytes, void **out) {
  if (!out) return SGTD_ERR_INVALID;
  *out = bytes ? malloc(bytes) : nullptr;
  return (*out || !bytes) ? SGTD_OK : SGTD_ERR_HIP;
}
int sgtd_host_free(void *p) { free(p); return SGTD_OK; }

}  // extern "C"
