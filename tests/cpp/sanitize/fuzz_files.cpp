// fuzz_files.cpp — mutation fuzzing, under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU, of everything in
// the library that parses bytes from disk (no GPU needed, none used):
//   1. the graph-JSON scanner (ingest::parse_graph on exact-size heap buffers without a terminator, so that a read past
//      the end is a report; and sgtd_graphs_load on files): byte flips, deletions, insertions, truncations, splices of
//      valid documents in the producer's format (get_json.cpp:332-341)
//   2. the binary graph cache (sgtd_graphs_load_cache): every truncation of a small cache, bit flips, tampered counts
//   3. the saved table's header (read_table_header, what sgtd_load_table validates before it sizes a device buffer)
// plus the hand-written cases of the two duplicate-key policies and of \u escapes (surrogates).
// Every call must return OK or an error — never crash, never read or write out of bounds, never loop; whatever is
// returned as OK must be self-consistent (offsets ascending, inside the arrays; all of it is read here).
//
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all fuzz_files.cpp ingest_host.cpp -pthread
#include <sys/stat.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/sgtd_accel.h"
#include "../../../sgtd_amd/csrc/graph_ingest.hip.h"
#include "../../../sgtd_amd/csrc/table_file.h"

namespace {

struct Rng {
  uint64_t s;
  uint64_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
  size_t below(size_t n) { return n ? (size_t)(next() % n) : 0; }
};

std::string valid_doc(Rng &r, int n) {
  std::string d = "{\"nodes\": [";
  for (int i = 0; i < n; i++) d += (i ? ", " : "") + std::to_string(3 + r.below(9));
  d += "], \"edges\": [[0.0, 1.0]], \"weights\": [0.5], \"centers\": [";
  char b[128];
  for (int i = 0; i < n; i++) {
    snprintf(b, sizeof b, "%s[%.17g, %.17g, %.9g]", i ? ", " : "", (double)r.below(100000) / 997.0 - 50.0, (double)r.below(100000) / 991.0 - 50.0, (double)r.below(3000) / 1000.0);
    d += b;
  }
  d += "], \"poses\": [1.0, 0.0, 0.0, 12.5, 0.0, 1.0, 0.0, -3.25, 0.0, 0.0, 1.0, 0.5], \"volumes\": [], \"densitys\": [], \"k\\u00e9y\": {\"a\": [true, false, null, \"x\\ud83d\\ude00\"]}}";
  return d;
}

std::string mutate(Rng &r, const std::string &in, const std::string &other) {
  std::string d = in;
  const int n_mut = 1 + (int)r.below(4);
  static const char kBytes[] = "{}[],:\"\\-+.eE0123456789 \n\ttruefalsnu\x00\xff\x80";
  for (int m = 0; m < n_mut && !d.empty(); m++) {
    const size_t at = r.below(d.size());
    switch (r.below(7)) {
      case 0: d[at] = (char)(d[at] ^ (1 << r.below(8))); break;
      case 1: d[at] = kBytes[r.below(sizeof(kBytes) - 1)]; break;
      case 2: d.erase(at, 1 + r.below(8)); break;
      case 3: d.insert(at, 1, kBytes[r.below(sizeof(kBytes) - 1)]); break;
      case 4: d.resize(at); break;
      case 5: d.insert(at, other.substr(r.below(other.size()), r.below(40))); break;
      default: d.insert(at, std::string(1 + r.below(300), "[{\""[r.below(3)])); break;     // deep nesting / runs of openers
    }
  }
  return d;
}

bool consistent(const ingest::OneGraph &g) { return g.label.size() * 3 == g.xyz.size(); }

int fail(const char *what) { std::printf("FAILED: %s\n", what); return 1; }

// parse from a heap block of exactly the document's size (no terminator behind it)
bool parse_exact(const std::string &doc, ingest::OneGraph &g, bool last_wins) {
  char *buf = (char *)malloc(doc.size() ? doc.size() : 1);
  memcpy(buf, doc.data(), doc.size());
  const bool ok = ingest::parse_graph(buf, doc.size(), g, last_wins);
  free(buf);
  return ok;
}

int hand_cases() {
  struct Case { const char *doc; bool ok_first, ok_last; int first_label, last_label; };
  const Case cases[] = {
      // a repeated key: 3.1.1 keeps the first value, 3.2+ the last
      {"{\"nodes\":[3],\"centers\":[[1,2,3]],\"poses\":[],\"nodes\":[7]}", true, true, 3, 7},
      // ... whose discarded occurrence may be of any type
      {"{\"nodes\":\"abc\",\"centers\":[[1,2,3]],\"poses\":[],\"nodes\":[7]}", false, true, 0, 7},
      {"{\"nodes\":[3],\"centers\":[[1,2,3]],\"poses\":[],\"nodes\":{\"x\":1}}", true, false, 3, 0},
      // the repeated key changes the lengths
      {"{\"nodes\":[3,4],\"centers\":[[1,2,3]],\"poses\":[],\"centers\":[[1,2,3],[4,5,6]]}", false, true, 0, 3},
      // keys spelled with escapes; surrogate pairs in strings that are skipped
      {"{\"no\\u0064es\":[5],\"centers\":[[1,2,3]],\"poses\":[1],\"s\":\"\\ud83d\\ude00\"}", true, true, 5, 5},
      // a high surrogate without its low one, a lone low surrogate: parse errors (as in nlohmann::json)
      {"{\"nodes\":[5],\"centers\":[[1,2,3]],\"poses\":[],\"s\":\"\\ud83dx\"}", false, false, 0, 0},
      {"{\"nodes\":[5],\"centers\":[[1,2,3]],\"poses\":[],\"s\":\"\\ud83d\\u0041\"}", false, false, 0, 0},
      {"{\"nodes\":[5],\"centers\":[[1,2,3]],\"poses\":[],\"s\":\"\\ude00\"}", false, false, 0, 0},
      {"{}", false, false, 0, 0},
      {"", false, false, 0, 0},
      {"{\"nodes\":[1],\"centers\":[[1,2]],\"poses\":[]}", false, false, 0, 0},
      {"{\"nodes\":[1],\"centers\":[[1,2,3]],\"poses\":[] tru", false, false, 0, 0},
  };
  for (const Case &c : cases) {
    for (int lw = 0; lw < 2; lw++) {
      ingest::OneGraph g;
      const bool ok = parse_exact(c.doc, g, lw != 0);
      const bool want = lw ? c.ok_last : c.ok_first;
      if (ok != want) { std::printf("case %s (last_wins %d): ok %d, want %d (%s)\n", c.doc, lw, ok, want, g.error.c_str()); return 1; }
      if (ok && (g.label.empty() || (int)g.label[0] != (lw ? c.last_label : c.first_label))) { std::printf("case %s (last_wins %d): label %d\n", c.doc, lw, g.label.empty() ? -1 : (int)g.label[0]); return 1; }
      if (ok && !consistent(g)) return fail("inconsistent hand case");
    }
  }
  return 0;
}

void write_file(const std::string &path, const void *p, size_t n) {
  FILE *f = fopen(path.c_str(), "wb");
  if (n) fwrite(p, 1, n, f);
  fclose(f);
}

// reads everything a batch hands out (ASan checks every byte); false if its offsets are not self-consistent
bool touch_batch(sgtd_graph_batch *b, double *sink) {
  int nf = 0;
  int64_t nk = 0;
  const float *xyz, *poses;
  const uint32_t *label;
  const int64_t *off;
  if (sgtd_graphs_view(b, &nf, &nk, &xyz, &label, &off, &poses) != SGTD_OK) return false;
  if (off[0] != 0 || off[nf] != nk) return false;
  for (int f = 0; f < nf; f++) {
    if (off[f + 1] < off[f]) return false;
    for (int64_t k = off[f]; k < off[f + 1]; k++) *sink += xyz[3 * k] + xyz[3 * k + 1] + xyz[3 * k + 2] + (double)label[k];
    for (int k = 0; k < 12; k++) *sink += poses[12 * f + k];
  }
  return true;
}

}  // namespace

int main(int argc, char **argv) {
  const int n_json = argc > 1 ? atoi(argv[1]) : 20000;
  const int n_bin = argc > 2 ? atoi(argv[2]) : 3000;
  std::string dir = argc > 3 ? argv[3] : "/tmp/sgtd_fuzz_files";
  mkdir(dir.c_str(), 0755);
  Rng r{argc > 4 ? strtoull(argv[4], nullptr, 0) : 0x1234567ull};      // (a fourth argument: another stream of mutations)
  double sink = 0;
  if (hand_cases()) return 1;

  // ---- 1. the JSON scanner
  long ok_docs = 0;
  for (int i = 0; i < n_json; i++) {
    const std::string a = valid_doc(r, 1 + (int)r.below(12)), b = valid_doc(r, 1 + (int)r.below(5));
    const std::string d = i % 50 == 0 ? a : mutate(r, a, b);
    for (int lw = 0; lw < 2; lw++) {
      ingest::OneGraph g;
      if (parse_exact(d, g, lw != 0)) {
        if (!consistent(g)) return fail("a parsed graph with labels and centers of different lengths");
        ok_docs++;
        for (float v : g.xyz) sink += v;
      } else if (g.error.empty()) {
        return fail("a refused document without a reason");
      }
    }
    if (i % 100 == 0) {      // the same through the file interface (threads, the batch arrays)
      const std::string p0 = dir + "/a.json", p1 = dir + "/b.json";
      write_file(p0, d.data(), d.size());
      write_file(p1, a.data(), a.size());
      const char *paths[3] = {p1.c_str(), p0.c_str(), p1.c_str()};
      sgtd_graph_batch *gb = nullptr;
      const int st = sgtd_graphs_load(paths, 3, 2, &gb);
      if (st == SGTD_OK) { if (!touch_batch(gb, &sink)) return fail("an inconsistent batch from sgtd_graphs_load"); }
      else if (st != SGTD_ERR_IO || !*sgtd_graphs_error(gb)) return fail("sgtd_graphs_load: another status than IO, or no message");
      sgtd_graphs_free(gb);
    }
  }

  // ---- 2. the binary cache
  std::vector<std::string> docs;
  for (int i = 0; i < 3; i++) { docs.push_back(dir + "/c" + std::to_string(i) + ".json"); const std::string d = valid_doc(r, 2 + i); write_file(docs.back(), d.data(), d.size()); }
  const char *dp[3] = {docs[0].c_str(), docs[1].c_str(), docs[2].c_str()};
  sgtd_graph_batch *gb = nullptr;
  if (sgtd_graphs_load(dp, 3, 1, &gb) != SGTD_OK) return fail("valid documents refused");
  const std::string cache = dir + "/batch.cache", bad = dir + "/bad.cache";
  if (sgtd_graphs_save_cache(gb, cache.c_str()) != SGTD_OK) return fail("cache not written");
  sgtd_graphs_free(gb);
  std::string bytes;
  if (!ingest::read_file(cache, bytes)) return fail("cache not readable");
  long ok_caches = 0;
  auto try_cache = [&](const std::string &b) -> int {
    write_file(bad, b.data(), b.size());
    sgtd_graph_batch *c = nullptr;
    const int st = sgtd_graphs_load_cache(bad.c_str(), &c);
    int rc = 0;
    if (st == SGTD_OK) { ok_caches++; if (!touch_batch(c, &sink)) rc = fail("an inconsistent batch from a damaged cache"); }
    else if (st != SGTD_ERR_IO) rc = fail("sgtd_graphs_load_cache: another status than IO");
    sgtd_graphs_free(c);
    return rc;
  };
  if (try_cache(bytes) || ok_caches != 1) return fail("the intact cache was refused");
  for (size_t n = 0; n < bytes.size(); n++)
    if (try_cache(bytes.substr(0, n))) return 1;                 // every truncation
  for (int i = 0; i < n_bin; i++) {
    std::string b = bytes;
    const int flips = 1 + (int)r.below(3);
    for (int k = 0; k < flips; k++) {
      const size_t at = (r.below(4) == 0) ? r.below(24 + 32) % b.size() : r.below(b.size());    // (header and offsets more often)
      b[at] = (char)(b[at] ^ (1 << r.below(8)));
    }
    if (r.below(8) == 0) b += std::string(r.below(64), 'x');
    if (try_cache(b)) return 1;
  }
  {   // counts that agree with the file size but not with the offsets inside
    std::string b = bytes;
    int64_t nf, nk;
    memcpy(&nf, &b[8], 8); memcpy(&nk, &b[16], 8);
    const int64_t nf2 = nf + 2, nk2 = nk - 6;      // (nf + 1) * 8 + nf * 48 + nk * 16: +112 - 96 ... keep the size by padding
    memcpy(&b[8], &nf2, 8); memcpy(&b[16], &nk2, 8);
    const long long want = 24 + (nf2 + 1) * 8 + nf2 * 48 + nk2 * 16;
    b.resize((size_t)want, '\x01');
    if (try_cache(b)) return 1;
  }

  // ---- 3. the saved table's header
  sgtd_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.descriptor_near_num = 10; cfg.candidate_num = 50; cfg.max_frame_n = 20000; cfg.descriptor_min_len = 0.5; cfg.descriptor_max_len = 50.0;
  cfg.std_side_resolution = 1.0; cfg.rough_dis_threshold = 0.03;
  TableHeader h{};
  h.side_resolution = 1.0; h.min_len = 0.5; h.max_len = 50.0; h.near_num = 10; h.have_frames = 1; h.current_frame_id = 3; h.frame_lo = 0; h.frame_hi = 2;
  h.n_entries = 5; h.n_add_calls = 3;
  std::string tb(kTableMagic, 8);
  tb.append((const char *)&h, sizeof h);
  tb.append((size_t)(h.n_entries * kTableEntryBytes), '\x02');
  const std::string tpath = dir + "/table.bin";
  long ok_tables = 0;
  auto try_table = [&](const std::string &b) -> int {
    write_file(tpath, b.data(), b.size());
    FILE *f = fopen(tpath.c_str(), "rb");
    TableHeader got{};
    std::string err;
    const int st = read_table_header(f, tpath.c_str(), cfg, got, err);
    int rc = 0;
    if (st == SGTD_OK) {
      ok_tables++;
      // what sgtd_load_table relies on afterwards
      if (got.n_entries < 0 || got.n_entries >= (1ll << 32) - 2 || (long long)b.size() != 8 + (long long)sizeof(TableHeader) + got.n_entries * kTableEntryBytes ||
          (got.have_frames && (got.frame_lo > got.frame_hi || got.frame_hi >= (uint32_t)cfg.max_frame_n)) || ftell(f) != 8 + (long)sizeof(TableHeader))
        rc = fail("an accepted table header that breaks what the loader relies on");
    } else if (err.empty()) {
      rc = fail("a refused table header without a reason");
    }
    fclose(f);
    return rc;
  };
  if (try_table(tb) || ok_tables != 1) return fail("the intact table header was refused");
  for (size_t n = 0; n < 8 + sizeof(TableHeader) + 4; n++)
    if (try_table(tb.substr(0, n))) return 1;
  for (int i = 0; i < n_bin; i++) {
    std::string b = tb;
    const int flips = 1 + (int)r.below(3);
    for (int k = 0; k < flips; k++) { const size_t at = r.below(8 + sizeof(TableHeader)); b[at] = (char)(b[at] ^ (1 << r.below(8))); }
    if (r.below(4) == 0) b.resize(r.below(b.size() + 64), 'y');
    if (try_table(b)) return 1;
  }
  std::printf("file fuzzing: ok (%d JSON documents x 2 policies, %ld accepted; %zu cache truncations + %d mutations, %ld accepted; %d table headers, %ld accepted; sink %g)\n",
              n_json, ok_docs, bytes.size(), n_bin, ok_caches, n_bin, ok_tables, sink);
  return 0;
}
