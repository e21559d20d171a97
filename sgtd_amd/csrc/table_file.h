// table_file.h — the header of a saved table (sgtd_save_table / sgtd_load_table) and its validation
// against the handle's configuration and the file's size, BEFORE anything is allocated from its counts.  Host code
// only: sgtd_accel.hip includes it, and so does the host-only sanitizer build (tests/cpp/sanitize/fuzz_files.cpp),
// which feeds it truncated and bit-flipped files.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>

#include "../../include/sgtd_accel.h"

static const char kTableMagic[8] = {'S', 'G', 'T', 'D', 'T', 'B', '0', '1'};
struct TableHeader {
  double side_resolution, min_len, max_len;
  int32_t near_num, have_frames;
  uint32_t current_frame_id, frame_lo, frame_hi, reserved;
  int64_t n_entries, n_add_calls;
};
// bytes of one table entry in the file: side, angle, center f64 x 3, vertex f32 x 9, label i32 x 3, frame u32, node_id i32 x 3
static const long long kTableEntryBytes = 3 * 3 * 8 + 9 * 4 + 3 * 4 + 4 + 3 * 4;

// Reads and checks the header of an open file (positioned at its start; left behind the header).  SGTD_OK, or the
// status sgtd_load_table returns, with `err` naming the reason.
inline int read_table_header(FILE *f, const char *path, const sgtd_config &cfg, TableHeader &h, std::string &err) {
  char magic[8];
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, kTableMagic, 8) != 0 || fread(&h, sizeof(h), 1, f) != 1 || h.n_entries < 0) {
    err = std::string(path) + ": not a saved table";
    return SGTD_ERR_IO;
  }
  if (!(h.side_resolution == cfg.std_side_resolution)) {      // (a NaN in a damaged header fails here too)
    err = "saved table was built with another std_side_resolution";
    return SGTD_ERR_INVALID;
  }
  if (h.near_num != cfg.descriptor_near_num || !(h.min_len == cfg.descriptor_min_len) || !(h.max_len == cfg.descriptor_max_len)) {
    err = "saved table was built with another descriptor_near_num / min_len / max_len";
    return SGTD_ERR_INVALID;
  }
  if (h.have_frames && (h.frame_hi >= (uint32_t)cfg.max_frame_n || h.frame_lo > h.frame_hi)) {
    err = std::string(path) + ": frame ids beyond max_frame_n";
    return SGTD_ERR_FRAME_LIMIT;
  }
  if (h.n_entries >= (1ll << 32) - 2) { err = std::string(path) + ": more entries than a 32-bit entry index holds"; return SGTD_ERR_UNSUPPORTED; }
  if ((h.n_entries > 0) != (h.have_frames != 0) || h.n_add_calls < 0) { err = std::string(path) + ": inconsistent table header"; return SGTD_ERR_IO; }
  // the entry count must agree with the file's size before any buffer is sized by it
  const long here = ftell(f);
  if (here < 0 || fseek(f, 0, SEEK_END) != 0) { err = std::string(path) + ": cannot seek"; return SGTD_ERR_IO; }
  const long long fsize = (long long)ftell(f);
  if (fseek(f, here, SEEK_SET) != 0) { err = std::string(path) + ": cannot seek"; return SGTD_ERR_IO; }
  if (fsize != (long long)here + h.n_entries * kTableEntryBytes) {
    err = std::string(path) + ": truncated or damaged table file";
    return SGTD_ERR_IO;
  }
  return SGTD_OK;
}
