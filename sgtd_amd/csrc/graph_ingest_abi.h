// graph_ingest_abi.h — the C ABI of the graph-JSON ingest (include/sgtd_accel.h, "graph-JSON ingest") over
// graph_ingest.hip.h.  Host code only: included once by sgtd_accel.hip, and once by the host-only sanitizer
// build (tests/cpp/sanitize/ingest_host.cpp: g++ -fsanitize=address,undefined, no HIP).
#pragma once
#include "../../include/sgtd_accel.h"
#include "graph_ingest.hip.h"

extern "C" {

int sgtd_graphs_load(const char *const *paths, int n_files, int n_threads, sgtd_graph_batch **out) {
  if (!out || n_files < 0 || (n_files > 0 && !paths)) return SGTD_ERR_INVALID;
  sgtd_graph_batch *b = new sgtd_graph_batch();
  *out = b;
  try {
    return ingest::load(paths, n_files, n_threads, *b) ? SGTD_OK : SGTD_ERR_IO;
  } catch (const std::exception &ex) {   // nothing throws across the ABI
    b->error = std::string("graph ingest: ") + ex.what();
    return SGTD_ERR_IO;
  }
}

int sgtd_graphs_save_cache(const sgtd_graph_batch *b, const char *path) {
  if (!b || !path) return SGTD_ERR_INVALID;
  return ingest::save_cache(*b, path) ? SGTD_OK : SGTD_ERR_IO;
}

int sgtd_graphs_load_cache(const char *path, sgtd_graph_batch **out) {
  if (!out || !path) return SGTD_ERR_INVALID;
  sgtd_graph_batch *b = new sgtd_graph_batch();
  *out = b;
  try {
    return ingest::load_cache(path, *b) ? SGTD_OK : SGTD_ERR_IO;
  } catch (const std::exception &ex) {
    b->error = std::string(path) + ": " + ex.what();
    return SGTD_ERR_IO;
  }
}

int sgtd_graphs_view(const sgtd_graph_batch *b, int *n_frames, int64_t *n_keypoints, const float **xyz,
                     const uint32_t **label, const int64_t **kp_off, const float **poses) {
  if (!b || b->kp_off.empty()) return SGTD_ERR_INVALID;
  if (n_frames) *n_frames = (int)b->kp_off.size() - 1;
  if (n_keypoints) *n_keypoints = (int64_t)b->label.size();
  if (xyz) *xyz = b->xyz.data();
  if (label) *label = b->label.data();
  if (kp_off) *kp_off = b->kp_off.data();
  if (poses) *poses = b->poses.data();
  return SGTD_OK;
}

const char *sgtd_graphs_error(const sgtd_graph_batch *b) { return b ? b->error.c_str() : ""; }

void sgtd_graphs_free(sgtd_graph_batch *b) { delete b; }

}  // extern "C"
