// Differential test of the graph-JSON ingest (sgtd_graphs_load, sgtd_amd/csrc/graph_ingest.hip.h) against the
// reference's own JSON library: every fuzzed document is parsed with nlohmann::json exactly as
// src/sgtd/include/Semantic_Graph.hpp:122-184 does — `inputFile >> j`, `j["nodes"].get<std::vector<int>>()`,
// `item[k].get<float>()` for every center, `j["poses"].get<std::vector<float>>()` — and labelled as
// include/utility.hpp:646-659 does (label[i] -> uint32), and compared BIT FOR BIT with what sgtd_graphs_load
// returns for the same files.  The header is the one this image carries (/opt/conda/include/json.hpp, 3.1.1);
// the reference's tree does not pin a version.
//
//   g++ -std=c++17 -O1 tests/cpp/test_ingest_nlohmann.cpp -Iinclude -Lsgtd_amd -lsgtd_accel ... ; ./a.out [n_docs] [dir]
#include "/opt/conda/include/json.hpp"

#include <sys/stat.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "sgtd_accel.h"

namespace {

struct Rng {
  uint64_t s;
  uint64_t next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return s >> 11; }
  int below(int n) { return (int)(next() % (uint64_t)n); }
  bool chance(int pct) { return below(100) < pct; }
};

const char *kSpecialFloat[] = {"-0", "-0.0", "0", "0.0", "-0e0", "0E+0", "11.0", "11", "16777217", "16777217.0", "2147483648", "4294967296",
                               "9007199254740993", "9007199254740993.0", "18446744073709551615", "18446744073709551616", "-9223372036854775808",
                               "-9223372036854775809", "1e-45", "1.5e-45", "1e-50", "3.4028234e38", "3.4028235e38", "1e38", "1E2", "1e+2", "2.5E-3",
                               "0.30000000000000004", "0.1", "123456.789012345678", "-7.0000001", "5e-324", "1.7976931348623157e308"};
const char *kSpecialInt[] = {"0", "-0", "3", "11", "11.0", "11.9", "-3.5", "255", "-1", "4294967301", "-4294967291", "2147483647", "-2147483648", "7e0", "1.2e1"};

std::string ws(Rng &r) {
  static const char *w[] = {"", "", "", " ", "\n", "\t", "  ", "\r\n", " \n  "};
  return w[r.below(9)];
}

std::string num_float(Rng &r) {
  switch (r.below(6)) {
    case 0: return kSpecialFloat[r.below((int)(sizeof(kSpecialFloat) / sizeof(*kSpecialFloat)))];
    case 1: return std::to_string((long long)r.below(200001) - 100000);
    case 2: { char b[64]; snprintf(b, sizeof b, "%.*f", r.below(17) + 1, ((double)r.next() / 9007199254740992.0 - 0.5) * 200.0); return b; }
    case 3: { char b[64]; snprintf(b, sizeof b, "%.*e", r.below(17), ((double)r.next() / 9007199254740992.0 - 0.5) * 1e3); return b; }
    case 4: { char b[64]; snprintf(b, sizeof b, "%.17g", ((double)r.next() / 9007199254740992.0 - 0.5) * 120.0); return b; }
    default: { char b[64]; snprintf(b, sizeof b, "%dE%s%d", r.below(2000) - 1000, r.chance(50) ? "-" : (r.chance(50) ? "+" : ""), r.below(5)); return b; }
  }
}

std::string num_int(Rng &r) {
  if (r.chance(25)) return kSpecialInt[r.below((int)(sizeof(kSpecialInt) / sizeof(*kSpecialInt)))];
  return std::to_string(r.below(20) - 2);
}

std::string junk_value(Rng &r, int depth) {
  switch (r.below(depth > 3 ? 4 : 7)) {
    case 0: return num_float(r);
    case 1: return "\"a }] \\\" \\\\ { [ \\u00e9 , :\"";
    case 2: return r.chance(50) ? "true" : "null";
    case 3: return "\"\"";
    case 4: { std::string s = "[" + ws(r); const int n = r.below(4); for (int i = 0; i < n; i++) s += (i ? "," : "") + ws(r) + junk_value(r, depth + 1) + ws(r); return s + "]"; }
    case 5: { std::string s = "{" + ws(r); const int n = r.below(3);
              for (int i = 0; i < n; i++) s += std::string(i ? "," : "") + ws(r) + (r.chance(30) ? "\"nodes\"" : r.chance(30) ? "\"centers\"" : "\"k" + std::to_string(i) + "\"") + ws(r) + ":" + ws(r) + junk_value(r, depth + 1) + ws(r);
              return s + "}"; }
    default: return "[[1,2],[3.5,4e1],[]]";
  }
}

std::string nodes_array(Rng &r, int n) {
  std::string s = "[" + ws(r);
  for (int i = 0; i < n; i++) s += (i ? "," : "") + ws(r) + num_int(r) + ws(r);
  return s + "]";
}
std::string centers_array(Rng &r, int n) {
  std::string s = "[" + ws(r);
  for (int i = 0; i < n; i++) {
    s += (i ? "," : "") + ws(r) + "[" + ws(r);
    const int m = r.chance(10) ? 4 + r.below(2) : 3;          // (a longer item: only the first three are read)
    for (int k = 0; k < m; k++) s += (k ? "," : "") + ws(r) + num_float(r) + ws(r);
    s += "]" + ws(r);
  }
  return s + "]";
}
std::string poses_array(Rng &r) {
  const int n = r.chance(85) ? 12 : r.below(16);
  std::string s = "[" + ws(r);
  for (int i = 0; i < n; i++) s += (i ? "," : "") + ws(r) + num_float(r) + ws(r);
  return s + "]";
}

std::string key_spelling(Rng &r, const char *k) {
  if (!r.chance(8)) return std::string("\"") + k + "\"";
  // the same key with one character written as a \u escape
  std::string s = "\"";
  const int at = r.below((int)strlen(k));
  for (int i = 0; k[i]; i++) {
    if (i == at) { char b[8]; snprintf(b, sizeof b, "\\u%04x", (unsigned)k[i]); s += b; }
    else s += k[i];
  }
  return s + "\"";
}

std::string document(Rng &r) {
  const int n = r.chance(5) ? 0 : 1 + r.below(40);
  struct Member { std::string key, value; };
  std::vector<Member> m;
  m.push_back({key_spelling(r, "nodes"), nodes_array(r, n)});
  m.push_back({key_spelling(r, "centers"), centers_array(r, n)});
  m.push_back({key_spelling(r, "poses"), poses_array(r)});
  static const char *extra[] = {"edges", "weights", "volumes", "densitys", "meta", "node", "Nodes", "centers2"};
  const int ne = r.below(6);
  for (int i = 0; i < ne; i++) m.push_back({std::string("\"") + extra[r.below(8)] + "\"", junk_value(r, 0)});
  if (r.chance(15)) {      // a key twice: different content the second time (same length, so that either choice is a valid graph)
    const int which = r.below(3);
    m.push_back({which == 0 ? "\"nodes\"" : which == 1 ? "\"centers\"" : "\"poses\"", which == 0 ? nodes_array(r, n) : which == 1 ? centers_array(r, n) : poses_array(r)});
  }
  for (size_t i = m.size(); i > 1; i--) std::swap(m[i - 1], m[(size_t)r.below((int)i)]);
  std::string s = ws(r) + "{" + ws(r);
  for (size_t i = 0; i < m.size(); i++) s += (i ? "," : "") + ws(r) + m[i].key + ws(r) + ":" + ws(r) + m[i].value + ws(r);
  return s + "}" + ws(r);
}

template <class T>
bool same_bits(const T *a, const T *b, size_t n) { return n == 0 || memcmp(a, b, n * sizeof(T)) == 0; }

}  // namespace

int main(int argc, char **argv) {
  // this image's json.hpp is 3.1.1: a repeated key keeps its FIRST value there (the library's default mirrors 3.2+: the last)
  setenv("SGTD_JSON_DUPLICATE_KEYS", "first", 1);
  const int n_docs = argc > 1 ? atoi(argv[1]) : 1500;
  std::string dir = argc > 2 ? argv[2] : "/tmp/sgtd_ingest_fuzz";
  mkdir(dir.c_str(), 0755);
  Rng r{20251121};
  std::vector<std::string> paths;
  for (int i = 0; i < n_docs; i++) {
    paths.push_back(dir + "/g" + std::to_string(i) + ".json");
    std::ofstream(paths.back()) << document(r);
  }
  // the reference's way
  std::vector<float> ref_xyz, ref_pose;
  std::vector<uint32_t> ref_label;
  std::vector<int64_t> ref_off{0};
  for (const auto &p : paths) {
    std::ifstream in(p);
    nlohmann::json jj;
    in >> jj;
    const nlohmann::json &j = jj;
    std::vector<int> nodes = j["nodes"].get<std::vector<int>>();
    std::vector<float> centers;
    for (const auto &item : j["centers"]) {
      centers.push_back(item[0].get<float>()); centers.push_back(item[1].get<float>()); centers.push_back(item[2].get<float>());
    }
    std::vector<float> poses = j["poses"].get<std::vector<float>>();
    if (nodes.size() != centers.size() / 3) { printf("generator error: %s\n", p.c_str()); return 2; }
    for (size_t i = 0; i < nodes.size(); i++) ref_label.push_back((uint32_t)nodes[i]);      // temp_pt.label = label[i] (utility.hpp:656)
    ref_xyz.insert(ref_xyz.end(), centers.begin(), centers.end());
    for (int k = 0; k < 12; k++) ref_pose.push_back(k < (int)poses.size() ? poses[k] : 0.f);   // (the node reads poses[3], [7], [11]; shorter arrays: zeros here)
    ref_off.push_back((int64_t)ref_label.size());
  }
  // the product's way
  std::vector<const char *> cp;
  for (const auto &p : paths) cp.push_back(p.c_str());
  sgtd_graph_batch *b = nullptr;
  const int st = sgtd_graphs_load(cp.data(), n_docs, 4, &b);
  if (st != SGTD_OK) { printf("sgtd_graphs_load failed: %s\n", b ? sgtd_graphs_error(b) : "?"); return 1; }
  int nf = 0; int64_t nk = 0;
  const float *xyz, *poses; const uint32_t *label; const int64_t *off;
  sgtd_graphs_view(b, &nf, &nk, &xyz, &label, &off, &poses);
  int bad = 0;
  if (nf != n_docs || nk != (int64_t)ref_label.size()) { printf("counts differ: %d frames %lld keypoints vs %d / %zu\n", nf, (long long)nk, n_docs, ref_label.size()); bad++; }
  else {
    for (int f = 0; f < nf; f++) {
      const int64_t a = off[f], e = off[f + 1];
      const bool ok = a == ref_off[f] && e == ref_off[f + 1] && same_bits(xyz + 3 * a, ref_xyz.data() + 3 * a, (size_t)(3 * (e - a))) &&
                      same_bits(label + a, ref_label.data() + a, (size_t)(e - a)) && same_bits(poses + 12 * f, ref_pose.data() + 12 * f, 12);
      if (!ok && bad++ < 5) printf("document %s differs\n", paths[f].c_str());
    }
  }
  sgtd_graphs_free(b);
  if (argc <= 2) { for (const auto &p : paths) unlink(p.c_str()); rmdir(dir.c_str()); }
  if (bad) { printf("%d of %d documents differ from nlohmann::json\n", bad, n_docs); return 1; }
  printf("ingest equals nlohmann::json %d.%d.%d on %d fuzzed documents (%zu keypoints)\n", NLOHMANN_JSON_VERSION_MAJOR, NLOHMANN_JSON_VERSION_MINOR,
         NLOHMANN_JSON_VERSION_PATCH, n_docs, ref_label.size());
  return 0;
}
