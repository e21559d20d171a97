// multi_impl.hip.h — implementation of multi.hip.h, included at the end of sgtd_accel.hip: the
// single-device entry points are defined above it; every shard is an ordinary engine.
#pragma once
namespace multi {

static inline Group *G(sgtd_engine *e) { return e->grp; }

#define MCHK(call)                     \
  do {                                 \
    int _r = (call);                   \
    if (_r != SGTD_OK) {               \
      e->err = "device shard: " + std::string(sgtd_last_error(c)); \
      return _r;                       \
    }                                  \
  } while (0)

int create(const sgtd_config *cfg, const int *device_ids, int n_dev, sgtd_handle *out) {
  if (!cfg || !out || !device_ids || n_dev < 1 || n_dev > 64) return SGTD_ERR_INVALID;
  *out = nullptr;
  sgtd_engine *e = new sgtd_engine();
  e->cfg = *cfg;
  e->grp = new Group();
  Group *g = e->grp;
  g->n = n_dev;
  g->current_frame_id = cfg->first_frame_id;
  if (cfg->first_frame_id % (SGTD_SHARD_BLOCK * (u32)n_dev) != 0) { delete g; delete e; return SGTD_ERR_INVALID; }
  for (int s = 0; s < n_dev; s++) {
    sgtd_config c = *cfg;
    c.device_id = device_ids[s];
    c.first_frame_id = local_of(cfg->first_frame_id, n_dev);
    sgtd_handle h = nullptr;
    const int st = sgtd_create(&c, &h);
    if (st != SGTD_OK) {
      for (sgtd_engine *d : g->dev) sgtd_destroy(d);
      delete g; delete e;
      return st;
    }
    g->dev.push_back(h);
  }
  *out = e;
  return SGTD_OK;
}

int destroy(sgtd_engine *e) {
  for (sgtd_engine *d : G(e)->dev) sgtd_destroy(d);
  delete e->grp;
  delete e;
  return SGTD_OK;
}

// BuildSingleScanSTD of one frame: any device will do; stamped with the GLOBAL frame id
int build(sgtd_engine *e, const float *xyz, const uint32_t *label, int n, sgtd_desc_soa *out, int64_t capacity, int64_t *n_out) {
  if (!out || !n_out || n < 0 || (n > 0 && (!xyz || !label))) return SGTD_ERR_INVALID;
  Group *g = G(e);
  sgtd_engine *c = g->dev[shard_of(g->current_frame_id, g->n)];
  MCHK(sgtd_build(c, xyz, label, n, out, capacity, n_out));
  if (out->frame)
    for (int64_t i = 0; i < *n_out; i++) out->frame[i] = g->current_frame_id;
  return SGTD_OK;
}

// AddSTDescs: the frame goes to the owner of the current global id; descriptors must carry
// that id (what BuildSingleScanSTD stamped, STDesc.cpp:305) — it becomes the owner's local id
int add(sgtd_engine *e, const sgtd_desc_soa *d, int64_t n) {
  if (n < 0 || (n > 0 && (!d || !d->side || !d->label || !d->frame))) return SGTD_ERR_INVALID;
  Group *g = G(e);
  const u32 gid = g->current_frame_id;
  if ((uint64_t)gid >= (uint64_t)e->cfg.max_frame_n) return SGTD_ERR_FRAME_LIMIT;
  const int s = shard_of(gid, g->n);
  sgtd_engine *c = g->dev[s];
  std::vector<uint32_t> lf((size_t)std::max<int64_t>(n, 1));
  for (int64_t i = 0; i < n; i++) {
    if (d->frame[i] != gid) { e->err = "multi-device tables take descriptors stamped with the current frame id"; return SGTD_ERR_UNSUPPORTED; }
    lf[(size_t)i] = local_of(gid, g->n);
  }
  if (n == 0) {   // an empty frame (d may be NULL): the owner only advances its frame counter
    MCHK(sgtd_add(c, nullptr, 0));
  } else {
    sgtd_desc_soa t = *d;
    t.frame = lf.data();
    MCHK(sgtd_add(c, &t, n));
  }
  g->current_frame_id++;
  g->batch_valid = false;
  return SGTD_OK;
}

int add_frames(sgtd_engine *e, const float *xyz, const uint32_t *label, const int64_t *kp_off, int n_frames, int device_ptrs) {
  if (n_frames < 0 || !kp_off) return SGTD_ERR_INVALID;
  if (n_frames == 0) return SGTD_OK;
  Group *g = G(e);
  if (device_ptrs) { e->err = "multi-device tables take host pointers"; return SGTD_ERR_UNSUPPORTED; }
  if ((uint64_t)g->current_frame_id + (uint64_t)n_frames > (uint64_t)e->cfg.max_frame_n) return SGTD_ERR_FRAME_LIMIT;
  int k = 0;
  while (k < n_frames) {   // runs of frames with one owner (to the end of its block)
    const u32 gid = g->current_frame_id;
    const int run = std::min<int>(n_frames - k, (int)(SGTD_SHARD_BLOCK - gid % SGTD_SHARD_BLOCK));
    sgtd_engine *c = g->dev[shard_of(gid, g->n)];
    MCHK(sgtd_add_frames(c, xyz, label, kp_off + k, run, 0));
    g->current_frame_id += (u32)run;
    k += run;
  }
  g->batch_valid = false;
  return SGTD_OK;
}

int finalize(sgtd_engine *e) {
  for (sgtd_engine *c : G(e)->dev) MCHK(sgtd_finalize(c));
  return SGTD_OK;
}

int query_frames(sgtd_engine *e, const float *xyz, const uint32_t *label, const int64_t *kp_off, int n_queries, int device_ptrs) {
  if (n_queries <= 0 || !kp_off || !xyz || !label) return SGTD_ERR_INVALID;
  Group *g = G(e);
  if (device_ptrs) { e->err = "multi-device tables take host pointers"; return SGTD_ERR_UNSUPPORTED; }
  // every device builds the query descriptors itself and sweeps its shard; the calls only
  // enqueue, so the devices run concurrently
  for (sgtd_engine *c : g->dev) MCHK(sgtd_query_frames(c, xyz, label, kp_off, n_queries, 0));
  g->nq = n_queries; g->batch_valid = true; g->merged = false; g->verified = false;
  return SGTD_OK;
}

int query_descs(sgtd_engine *e, const sgtd_desc_soa *q, int64_t nq) {
  if (nq < 0 || (nq > 0 && (!q || !q->side || !q->label || !q->frame))) return SGTD_ERR_INVALID;
  Group *g = G(e);
  std::vector<uint32_t> lf((size_t)std::max<int64_t>(nq, 1));
  for (int s = 0; s < g->n; s++) {
    sgtd_engine *c = g->dev[s];
    // the frame test (:373) compares ids: a query id that names a map frame of this shard
    // becomes that frame's local id, any other id a value no local frame has
    for (int64_t i = 0; i < nq; i++) {
      const u32 f = q->frame[i];
      lf[(size_t)i] = (f < g->current_frame_id && shard_of(f, g->n) == s) ? local_of(f, g->n) : 0xFFFFFFFEu;
    }
    sgtd_desc_soa t = *q;
    t.frame = lf.data();
    MCHK(sgtd_query_descs(c, &t, nq));
  }
  g->nq = 1; g->batch_valid = true; g->merged = false; g->verified = false;
  return SGTD_OK;
}

// gathers the local candidate tables and merges them (STDesc.cpp:423-433 on the union)
int merge(sgtd_engine *e) {
  Group *g = G(e);
  if (!g->batch_valid) return SGTD_ERR_STATE;
  if (g->merged) return SGTD_OK;
  const int cn = e->cfg.candidate_num, nq = g->nq, n = g->n;
  std::vector<std::vector<int>> nc(n), cf(n), cv(n);
  std::vector<std::vector<int64_t>> po(n);
  for (int s = 0; s < n; s++) {
    sgtd_engine *c = g->dev[s];
    nc[s].resize(nq); cf[s].resize((size_t)nq * cn); cv[s].resize((size_t)nq * cn); po[s].resize((size_t)nq * (cn + 1));
    MCHK(sgtd_result_candidates(c, nc[s].data(), cf[s].data(), cv[s].data(), po[s].data()));
  }
  g->n_cand.assign(nq, 0);
  g->cand_frame.assign((size_t)nq * cn, -1); g->cand_votes.assign((size_t)nq * cn, 0);
  g->owner.assign((size_t)nq * cn, -1); g->owner_slot.assign((size_t)nq * cn, -1);
  g->pair_off.assign((size_t)nq * (cn + 1), 0);
  struct Item { long long votes; u32 frame; int s, k; };
  std::vector<Item> items;
  for (int q = 0; q < nq; q++) {
    items.clear();
    for (int s = 0; s < n; s++)
      for (int k = 0; k < nc[s][q]; k++) {
        const size_t i = (size_t)q * cn + k;
        if (cv[s][i] >= 5) items.push_back({cv[s][i], global_of(s, (u32)cf[s][i], n), s, k});
      }
    std::sort(items.begin(), items.end(), [](const Item &a, const Item &b) {
      return a.votes != b.votes ? a.votes > b.votes : a.frame < b.frame;
    });
    const int m = (int)std::min<size_t>(items.size(), (size_t)cn);
    g->n_cand[q] = m;
    long long acc = 0;
    for (int k = 0; k <= cn; k++) {
      g->pair_off[(size_t)q * (cn + 1) + k] = acc;
      if (k < m) {
        const Item &it = items[k];
        const size_t i = (size_t)q * cn + k;
        g->cand_frame[i] = (int)it.frame; g->cand_votes[i] = (int)it.votes; g->owner[i] = it.s; g->owner_slot[i] = it.k;
        acc += po[it.s][(size_t)q * (cn + 1) + it.k + 1] - po[it.s][(size_t)q * (cn + 1) + it.k];
      }
    }
  }
  g->merged = true;
  return SGTD_OK;
}

int result_candidates(sgtd_engine *e, int32_t *n_cand, int32_t *cand_frame, int32_t *cand_votes, int64_t *pair_off) {
  CHK(merge(e));
  Group *g = G(e);
  const int cn = e->cfg.candidate_num, nq = g->nq;
  if (n_cand) std::memcpy(n_cand, g->n_cand.data(), nq * sizeof(int));
  if (cand_frame) std::memcpy(cand_frame, g->cand_frame.data(), (size_t)nq * cn * sizeof(int));
  if (cand_votes) std::memcpy(cand_votes, g->cand_votes.data(), (size_t)nq * cn * sizeof(int));
  if (pair_off)
    for (size_t i = 0; i < (size_t)nq * (cn + 1); i++) pair_off[i] = g->pair_off[i];
  return SGTD_OK;
}

// match lists of query q in the merged candidate order, each fetched from its owner
int result_pairs(sgtd_engine *e, int q, int32_t *q_idx, int64_t *db_entry, int64_t capacity, int64_t *n_pairs) {
  if (!n_pairs) return SGTD_ERR_INVALID;
  CHK(merge(e));
  Group *g = G(e);
  if (q < 0 || q >= g->nq) return SGTD_ERR_INVALID;
  const int cn = e->cfg.candidate_num;
  const int64_t total = g->pair_off[(size_t)q * (cn + 1) + cn];
  *n_pairs = total;
  if (total > capacity) return SGTD_ERR_CAPACITY;
  std::vector<std::vector<int32_t>> qi(g->n);
  std::vector<std::vector<int64_t>> en(g->n), off(g->n);
  std::vector<bool> have(g->n, false);
  for (int k = 0; k < g->n_cand[q]; k++) {
    const size_t i = (size_t)q * cn + k;
    const int s = g->owner[i], ks = g->owner_slot[i];
    sgtd_engine *c = g->dev[s];
    if (!have[s]) {   // the owner's lists of this query, once
      off[s].resize((size_t)g->nq * (cn + 1));
      MCHK(sgtd_result_candidates(c, nullptr, nullptr, nullptr, off[s].data()));
      const int64_t ts = off[s][(size_t)q * (cn + 1) + cn];
      qi[s].resize((size_t)std::max<int64_t>(ts, 1)); en[s].resize((size_t)std::max<int64_t>(ts, 1));
      int64_t got = 0;
      MCHK(sgtd_result_pairs(c, q, qi[s].data(), en[s].data(), ts, &got));
      have[s] = true;
    }
    const int64_t lo = off[s][(size_t)q * (cn + 1) + ks], hi = off[s][(size_t)q * (cn + 1) + ks + 1];
    int64_t o = g->pair_off[i + (size_t)q];   // = pair_off[q * (cn + 1) + k]
    for (int64_t r = lo; r < hi; r++, o++) {
      if (q_idx) q_idx[o] = qi[s][(size_t)r];
      if (db_entry) db_entry[o] = ((int64_t)s << SGTD_ENTRY_SHARD_SHIFT) | en[s][(size_t)r];
    }
  }
  return SGTD_OK;
}

int fetch_entries(sgtd_engine *e, const int64_t *db_entry, int64_t n, sgtd_desc_soa *out) {
  if (n < 0 || (n > 0 && (!db_entry || !out))) return SGTD_ERR_INVALID;
  Group *g = G(e);
  // runs of one owner are fetched together; frame ids come back global
  int64_t i = 0;
  std::vector<int64_t> loc;
  while (i < n) {
    const int s = (int)(db_entry[i] >> SGTD_ENTRY_SHARD_SHIFT);
    if (s < 0 || s >= g->n) return SGTD_ERR_INVALID;
    int64_t j = i;
    loc.clear();
    while (j < n && (int)(db_entry[j] >> SGTD_ENTRY_SHARD_SHIFT) == s) { loc.push_back(db_entry[j] & ((1ll << SGTD_ENTRY_SHARD_SHIFT) - 1)); j++; }
    sgtd_desc_soa t;
    t.side = out->side ? out->side + 3 * i : nullptr; t.angle = out->angle ? out->angle + 3 * i : nullptr;
    t.center = out->center ? out->center + 3 * i : nullptr; t.vertex = out->vertex ? out->vertex + 9 * i : nullptr;
    t.label = out->label ? out->label + 3 * i : nullptr; t.frame = out->frame ? out->frame + i : nullptr;
    t.node_id = out->node_id ? out->node_id + 3 * i : nullptr;
    sgtd_engine *c = g->dev[s];
    MCHK(sgtd_fetch_entries(c, loc.data(), j - i, &t));
    if (t.frame)
      for (int64_t k = 0; k < j - i; k++) t.frame[k] = global_of(s, t.frame[k], g->n);
    i = j;
  }
  return SGTD_OK;
}

int result_query_desc_count(sgtd_engine *e, int q, int64_t *n) { sgtd_engine *c = G(e)->dev[0]; MCHK(sgtd_result_query_desc_count(c, q, n)); return SGTD_OK; }

int result_query_descs(sgtd_engine *e, int q, sgtd_desc_soa *out, int64_t capacity, int64_t *n_out) {
  Group *g = G(e);
  sgtd_engine *c = g->dev[0];
  MCHK(sgtd_result_query_descs(c, q, out, capacity, n_out));
  if (out->frame)
    for (int64_t i = 0; i < *n_out; i++) out->frame[i] = g->current_frame_id;   // :592: every query descriptor carries F
  return SGTD_OK;
}

int result_votes(sgtd_engine *e, int q, uint32_t *votes, int64_t capacity, uint32_t *frame_lo, int64_t *n) {
  Group *g = G(e);
  if (!n || !g->batch_valid || q < 0 || q >= g->nq) return SGTD_ERR_INVALID;
  const u32 lo = e->cfg.first_frame_id;
  const int64_t span = std::max<int64_t>((int64_t)g->current_frame_id - lo, 1);
  *n = span;
  if (frame_lo) *frame_lo = lo;
  if (!votes) return SGTD_OK;
  if (span > capacity) return SGTD_ERR_CAPACITY;
  std::fill(votes, votes + span, 0u);
  std::vector<uint32_t> lv;
  for (int s = 0; s < g->n; s++) {
    sgtd_engine *c = g->dev[s];
    uint32_t llo = 0; int64_t ln = 0;
    MCHK(sgtd_result_votes(c, q, nullptr, 0, &llo, &ln));
    lv.assign((size_t)ln, 0u);
    MCHK(sgtd_result_votes(c, q, lv.data(), ln, &llo, &ln));
    for (int64_t k = 0; k < ln; k++) {
      const u32 gf = global_of(s, llo + (u32)k, g->n);
      if (lv[(size_t)k] && gf >= lo && gf - lo < (u32)span) votes[gf - lo] = lv[(size_t)k];
    }
  }
  return SGTD_OK;
}

int verify(sgtd_engine *e) {
  Group *g = G(e);
  CHK(merge(e));
  for (sgtd_engine *c : g->dev) MCHK(sgtd_verify(c));   // every owner verifies its local candidates
  g->verified = true;
  return SGTD_OK;
}

int result_verify(sgtd_engine *e, int q, double *score, double *pose) {
  Group *g = G(e);
  if (!g->verified || q < 0 || q >= g->nq) return SGTD_ERR_INVALID;
  const int cn = e->cfg.candidate_num;
  std::vector<std::vector<double>> ss(g->n), pp(g->n);
  for (int k = 0; k < cn; k++) {
    if (score) score[k] = -1.0;
    if (pose) std::fill(pose + (size_t)k * 12, pose + (size_t)k * 12 + 12, 0.0);
  }
  for (int k = 0; k < g->n_cand[q]; k++) {
    const size_t i = (size_t)q * cn + k;
    const int s = g->owner[i], ks = g->owner_slot[i];
    sgtd_engine *c = g->dev[s];
    if (ss[s].empty()) {
      ss[s].resize(cn); pp[s].resize((size_t)cn * 12);
      MCHK(sgtd_result_verify(c, q, ss[s].data(), pp[s].data()));
    }
    if (score) score[k] = ss[s][ks];
    if (pose) std::copy(pp[s].begin() + (size_t)ks * 12, pp[s].begin() + (size_t)ks * 12 + 12, pose + (size_t)k * 12);
  }
  return SGTD_OK;
}

int result_inliers(sgtd_engine *e, int q, int cand, int32_t *idx, int64_t capacity, int64_t *n) {
  Group *g = G(e);
  if (!g->verified || q < 0 || q >= g->nq || cand < 0 || cand >= g->n_cand[q]) return SGTD_ERR_INVALID;
  const size_t i = (size_t)q * e->cfg.candidate_num + cand;
  sgtd_engine *c = g->dev[g->owner[i]];
  MCHK(sgtd_result_inliers(c, q, g->owner_slot[i], idx, capacity, n));
  return SGTD_OK;
}

// inlier pairs of every merged candidate of query q: each candidate's from its owner, entry ids with the owner in the upper bits
int result_inlier_pairs(sgtd_engine *e, int q, int64_t *cand_off, int32_t *q_idx, int64_t *db_entry, int64_t capacity, int64_t *n_pairs) {
  Group *g = G(e);
  if (!n_pairs || !cand_off || !g->verified || q < 0 || q >= g->nq) return SGTD_ERR_INVALID;
  const int cn = e->cfg.candidate_num;
  std::vector<std::vector<int32_t>> qi(g->n);
  std::vector<std::vector<int64_t>> en(g->n), off(g->n);
  std::vector<bool> have(g->n, false);
  int64_t o = 0;
  bool fits = true;
  for (int k = 0; k <= cn; k++) cand_off[k] = 0;
  for (int k = 0; k < g->n_cand[q]; k++) {
    const size_t i = (size_t)q * cn + k;
    const int s = g->owner[i], ks = g->owner_slot[i];
    if (!have[s]) {   // the owner's inlier pairs of this query, once
      sgtd_engine *c = g->dev[s];
      off[s].resize((size_t)cn + 1);
      int64_t ns = 0;
      int st = sgtd_result_inlier_pairs(c, q, off[s].data(), nullptr, nullptr, 0, &ns);
      if (st != SGTD_OK && st != SGTD_ERR_CAPACITY) MCHK(st);
      qi[s].resize((size_t)std::max<int64_t>(ns, 1)); en[s].resize((size_t)std::max<int64_t>(ns, 1));
      MCHK(sgtd_result_inlier_pairs(c, q, off[s].data(), qi[s].data(), en[s].data(), ns, &ns));
      have[s] = true;
    }
    cand_off[k] = o;
    for (int64_t r = off[s][(size_t)ks]; r < off[s][(size_t)ks + 1]; r++, o++) {
      if (o < capacity) {
        if (q_idx) q_idx[o] = qi[s][(size_t)r];
        if (db_entry) db_entry[o] = ((int64_t)s << SGTD_ENTRY_SHARD_SHIFT) | en[s][(size_t)r];
      } else fits = false;
    }
  }
  for (int k = g->n_cand[q]; k <= cn; k++) cand_off[k] = o;
  *n_pairs = o;
  return fits ? SGTD_OK : SGTD_ERR_CAPACITY;
}

// SearchLoop's choice on the merged list (STDesc.cpp:105-146)
int search_loop(sgtd_engine *e, double icp_threshold, int32_t *best_cand, int32_t *best_frame, double *best_score) {
  Group *g = G(e);
  if (!g->verified) return SGTD_ERR_INVALID;
  const int cn = e->cfg.candidate_num;
  std::vector<double> score(cn);
  for (int q = 0; q < g->nq; q++) {
    CHK(result_verify(e, q, score.data(), nullptr));
    double best = 0; int arg = -1;
    for (int k = 0; k < g->n_cand[q]; k++)
      if (score[k] > best) { best = score[k]; arg = k; }
    const bool ok = arg >= 0 && best > icp_threshold;
    if (best_cand) best_cand[q] = ok ? arg : -1;
    if (best_frame) best_frame[q] = ok ? g->cand_frame[(size_t)q * cn + arg] : -1;
    if (best_score) best_score[q] = ok ? best : 0.0;
  }
  return SGTD_OK;
}

int sync(sgtd_engine *e) {
  for (sgtd_engine *c : G(e)->dev) MCHK(sgtd_sync(c));
  return SGTD_OK;
}

int get_stats(sgtd_engine *e, sgtd_stats *out) {
  Group *g = G(e);
  sgtd_stats t{};
  for (int s = 0; s < g->n; s++) {
    sgtd_engine *c = g->dev[s];
    sgtd_stats x;
    MCHK(sgtd_get_stats(c, &x));
    t.n_entries += x.n_entries; t.n_buckets += x.n_buckets; t.n_frames += x.n_frames;
    t.last_queries = x.last_queries; t.last_D = x.last_D;
    t.last_P += x.last_P; t.last_M += x.last_M; t.last_P_swept += x.last_P_swept;
    t.hbm_bytes_table += x.hbm_bytes_table;
    t.overflowed |= x.overflowed;
    t.select_form = x.select_form;
    t.batches_total = x.batches_total; t.overflow_launches_total += x.overflow_launches_total;
    t.reruns_total += x.reruns_total; t.rewrites_total += x.rewrites_total;
    t.list_moves_total += x.list_moves_total; t.last_list_moves += x.last_list_moves;
    t.ms_total = std::max(t.ms_total, x.ms_total);
  }
  if (g->merged) {
    const int cn = e->cfg.candidate_num;
    for (int q = 0; q < g->nq; q++) t.last_cand_pairs += g->pair_off[(size_t)q * (cn + 1) + cn];
  }
  *out = t;
  return SGTD_OK;
}

int max_batch(sgtd_engine *e, int n_keypoints, int64_t *max_queries) {
  int64_t best = -1;
  for (sgtd_engine *c : G(e)->dev) {
    int64_t m = 0;
    MCHK(sgtd_max_batch(c, n_keypoints, &m));
    best = best < 0 ? m : std::min(best, m);
  }
  *max_queries = best;
  return SGTD_OK;
}

u32 current_frame_id(sgtd_engine *e) { return G(e)->current_frame_id; }
int device_count(sgtd_engine *e) { return G(e)->n; }
sgtd_engine *device_handle(sgtd_engine *e, int k) { return (k >= 0 && k < G(e)->n) ? G(e)->dev[k] : nullptr; }

#undef MCHK
}  // namespace multi
