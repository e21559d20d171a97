// graph_ingest.hip.h — host side: the reference's on-disk graph format -> keypoint batches
// (SURVEY §8f row 2).  Replaces readGraphFromFile / fromJSON
// (src/sgtd/include/Semantic_Graph.hpp:122-184) + Graph2CloudL (include/utility.hpp:646-659)
// for whole directories at once: files are parsed on host threads straight into the SoA
// layout sgtd_add_frames / sgtd_query_frames take (xyz f32, label u32, CSR offsets), plus the
// 12-float pose row of every frame (the node reads poses[3], [7], [11],
// semantic_graph_localization.cpp:447).
//
// Format (producer: src/get_json.cpp:332-341, Graph::toJSON Semantic_Graph.hpp:79-110):
//   {"nodes":[int...], "edges":[[..]], "weights":[..], "centers":[[x,y,z]...],
//    "poses":[12 floats], "volumes":[..], "densitys":[..]}
// Only "nodes", "centers", "poses" are read (fromJSON :157-164); every other key is skipped.
// Numbers as nlohmann::json (3.1.1, the header this image carries; tests/cpp/test_ingest_nlohmann.cpp
// compares with it bit for bit) reads them: a token without fraction and exponent is an integer —
// strtoull for a non-negative one, strtoll for a negative one ("-0" is the integer 0: +0.0f, not
// -0.0f), and only if that overflows a double — every other token goes through strtod; get<float>() /
// get<int>() are static_casts of whichever of the three was stored.  A key that occurs twice: nlohmann 3.1.1
// keeps the FIRST value (its parser emplaces), 3.2 and later keep the LAST (object[key] is assigned; ROS noetic
// ships 3.7.3) — SGTD_JSON_DUPLICATE_KEYS = first | last chooses, default last; the producer,
// get_json.cpp:332-341, never writes a key twice.
#pragma once
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <string>
#include <thread>
#include <vector>

struct sgtd_graph_batch {
  std::vector<float> xyz;        // [n_keypoints * 3]
  std::vector<uint32_t> label;   // [n_keypoints]
  std::vector<int64_t> kp_off;   // [n_frames + 1]
  std::vector<float> poses;      // [n_frames * 12]
  std::string error;
};

namespace ingest {

struct OneGraph {
  std::vector<float> xyz;
  std::vector<uint32_t> label;
  float pose[12];
  int n_pose = 0;
  std::string error;
};

// minimal JSON scanner: enough to walk any valid document and pull three arrays out
struct Scanner {
  const char *p, *end;
  std::string err;
  bool fail(const char *what) {
    if (err.empty()) err = what;
    return false;
  }
  void ws() {
    while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) p++;
  }
  bool expect(char c) {
    ws();
    if (p < end && *p == c) { p++; return true; }
    return fail("unexpected character");
  }
  bool peek(char c) {
    ws();
    return p < end && *p == c;
  }
  static void utf8(std::string *out, unsigned cp) {
    if (cp < 0x80) out->push_back((char)cp);
    else if (cp < 0x800) { out->push_back((char)(0xC0 | (cp >> 6))); out->push_back((char)(0x80 | (cp & 0x3F))); }
    else if (cp < 0x10000) { out->push_back((char)(0xE0 | (cp >> 12))); out->push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out->push_back((char)(0x80 | (cp & 0x3F))); }
    else { out->push_back((char)(0xF0 | (cp >> 18))); out->push_back((char)(0x80 | ((cp >> 12) & 0x3F))); out->push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out->push_back((char)(0x80 | (cp & 0x3F))); }
  }
  bool hex4(unsigned *cp) {
    if (p + 4 > end) return fail("bad \\u escape");
    unsigned v = 0;
    for (int k = 0; k < 4; k++) {
      const char c = p[k];
      const int d = c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1;
      if (d < 0) return fail("bad \\u escape");
      v = v * 16 + (unsigned)d;
    }
    p += 4;
    *cp = v;
    return true;
  }
  // a string with its escapes decoded (a key may be spelled "no\u0064es": the reference's parser compares decoded keys)
  bool string(std::string *out) {
    ws();
    if (p >= end || *p != '"') return fail("string expected");
    p++;
    while (p < end && *p != '"') {
      if (*p == '\\') {
        if (p + 1 >= end) return fail("bad escape");
        const char c = p[1];
        p += 2;
        if (c == 'u') {
          unsigned cp;
          if (!hex4(&cp)) return false;
          if (cp >= 0xD800 && cp < 0xDC00) {     // a high surrogate: \uDC00..\uDFFF must follow (nlohmann: parse error otherwise)
            if (!(p + 1 < end && p[0] == '\\' && p[1] == 'u')) return fail("a high surrogate without its low surrogate");
            p += 2;
            unsigned lo;
            if (!hex4(&lo)) return false;
            if (lo < 0xDC00 || lo > 0xDFFF) return fail("a high surrogate without its low surrogate");
            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
          } else if (cp >= 0xDC00 && cp <= 0xDFFF) {
            return fail("a low surrogate without its high surrogate");
          }
          if (out) utf8(out, cp);
        } else if (out) {
          out->push_back(c == 'n' ? '\n' : c == 't' ? '\t' : c == 'r' ? '\r' : c == 'b' ? '\b' : c == 'f' ? '\f' : c);   // (", \\, / stand for themselves)
        }
      } else {
        if (out) out->push_back(*p);
        p++;
      }
    }
    if (p >= end) return fail("unterminated string");
    p++;
    return true;
  }
  // a JSON number as nlohmann::json stores it: unsigned, signed or floating
  struct Num {
    int kind = 2;            // 0 number_unsigned, 1 number_integer, 2 number_float
    uint64_t u = 0;
    int64_t i = 0;
    double d = 0.0;
    float as_float() const { return kind == 0 ? static_cast<float>(u) : kind == 1 ? static_cast<float>(i) : static_cast<float>(d); }
    int as_int() const { return kind == 0 ? static_cast<int>(u) : kind == 1 ? static_cast<int>(i) : static_cast<int>(d); }
  };
  // The double strtod returns, without calling it, for tokens of at most 19 significant digits and a decimal exponent of
  // at most 27 in magnitude (every number the producer writes: floats dumped with 17 digits): the digits as an integer
  // (exact in the 64-bit significand of x87 extended precision) times or divided by a power of ten (exact there too:
  // 5^27 < 2^63) is ONE correctly rounded extended operation; rounding that to double gives the correctly rounded
  // double unless the extended result sits exactly on a midpoint of the double grid (low eleven significand bits
  // 100 0000 0000: the exact value decides) — then, and for every other token, strtod it is.
  // tests/cpp/test_ingest_numbers.cpp compares the two on millions of tokens.
  static bool fast_double(uint64_t m, int e10, bool neg, double *out) {
    static const long double p10[28] = {1e0L,  1e1L,  1e2L,  1e3L,  1e4L,  1e5L,  1e6L,  1e7L,  1e8L,  1e9L,  1e10L, 1e11L, 1e12L, 1e13L,
                                        1e14L, 1e15L, 1e16L, 1e17L, 1e18L, 1e19L, 1e20L, 1e21L, 1e22L, 1e23L, 1e24L, 1e25L, 1e26L, 1e27L};
    if (__LDBL_MANT_DIG__ != 64) return false;    // (no x87 extended precision here — e.g. this header's device pass: strtod)
    if (e10 < -27 || e10 > 27) return false;
    if (m == 0) { *out = neg ? -0.0 : 0.0; return true; }
    const long double x = e10 < 0 ? static_cast<long double>(m) / p10[-e10] : static_cast<long double>(m) * p10[e10];
    uint64_t sig;
    memcpy(&sig, &x, 8);                          // the 64-bit significand (explicit leading one)
    if ((sig & 0x7FFu) == 0x400u) return false;
    const double d = static_cast<double>(x);
    *out = neg ? -d : d;
    return true;
  }
  bool number(Num *out) {
    ws();
    if (p >= end) return fail("number expected");
    // the token by JSON's grammar: -? (0 | [1-9][0-9]*) (. [0-9]+)? ([eE] [+-]? [0-9]+)?
    // — and, on the way, its significant digits as an integer (the first 19) and its decimal exponent
    const char *q = p;
    bool is_float = false;
    const bool neg = *q == '-';
    if (neg) q++;
    if (q >= end || *q < '0' || *q > '9') return fail("number expected");
    uint64_t m = 0;
    int nd = 0, e10 = 0;
    bool all_digits = true;      // every significant digit is in m
    if (*q == '0') q++;
    else
      for (; q < end && *q >= '0' && *q <= '9'; q++) {
        if (nd < 19) { m = m * 10 + (uint64_t)(*q - '0'); nd++; }
        else { all_digits = false; e10++; }
      }
    if (q < end && *q == '.') {
      q++;
      if (q >= end || *q < '0' || *q > '9') return fail("digit expected after the decimal point");
      for (; q < end && *q >= '0' && *q <= '9'; q++) {
        if (nd == 0 && *q == '0') { e10--; continue; }     // zeros in front of the first significant digit
        if (nd < 19) { m = m * 10 + (uint64_t)(*q - '0'); nd++; e10--; }
        else all_digits = false;
      }
      is_float = true;
    }
    if (q < end && (*q == 'e' || *q == 'E')) {
      q++;
      bool eneg = false;
      if (q < end && (*q == '+' || *q == '-')) { eneg = *q == '-'; q++; }
      if (q >= end || *q < '0' || *q > '9') return fail("digit expected in the exponent");
      int ex = 0;
      for (; q < end && *q >= '0' && *q <= '9'; q++)
        if (ex < 100000) ex = ex * 10 + (*q - '0');
      e10 += eneg ? -ex : ex;
      is_float = true;
    }
    const char *tok_begin = p;
    p = q;
    if (!out) return true;
    Num n;
    if (!is_float) {
      // (no fraction, no exponent: all_digits means at most 19 digits — below 2^64; 18 fit a negative 64-bit integer)
      if (!neg && all_digits) { n.kind = 0; n.u = m; *out = n; return true; }
      if (neg && nd <= 18) { n.kind = 1; n.i = -static_cast<int64_t>(m); *out = n; return true; }
      const std::string tok(tok_begin, q);
      errno = 0;
      char *e = nullptr;
      if (neg) {
        const long long x = strtoll(tok.c_str(), &e, 10);
        if (errno == 0) { n.kind = 1; n.i = x; *out = n; return true; }
      } else {
        const unsigned long long x = strtoull(tok.c_str(), &e, 10);
        if (errno == 0) { n.kind = 0; n.u = x; *out = n; return true; }
      }
    }
    n.kind = 2;
    if (!(is_float && all_digits && fast_double(m, e10, neg, &n.d))) n.d = strtod(std::string(tok_begin, q).c_str(), nullptr);
    *out = n;
    return true;
  }
  // nesting is bounded (nlohmann's recursive-descent parser has no such bound, a hostile
  // file would overflow its stack; here it is a parse error)
  bool skip_value(int depth = 0) {
    ws();
    if (p >= end) return fail("value expected");
    if (depth > 256) return fail("nesting too deep");
    const char c = *p;
    if (c == '"') return string(nullptr);
    if (c == '{') {
      p++;
      if (peek('}')) { p++; return true; }
      while (true) {
        if (!string(nullptr) || !expect(':') || !skip_value(depth + 1)) return false;
        if (peek(',')) { p++; continue; }
        return expect('}');
      }
    }
    if (c == '[') {
      p++;
      if (peek(']')) { p++; return true; }
      while (true) {
        if (!skip_value(depth + 1)) return false;
        if (peek(',')) { p++; continue; }
        return expect(']');
      }
    }
    // (bounded by `end`: the buffer need not be NUL-terminated)
    if (end - p >= 4 && !memcmp(p, "true", 4)) { p += 4; return true; }
    if (end - p >= 5 && !memcmp(p, "false", 5)) { p += 5; return true; }
    if (end - p >= 4 && !memcmp(p, "null", 4)) { p += 4; return true; }
    return number(nullptr);
  }
  template <class F>
  bool array(F &&item) {   // [ item, item, ... ]
    if (!expect('[')) return false;
    if (peek(']')) { p++; return true; }
    while (true) {
      if (!item()) return false;
      if (peek(',')) { p++; continue; }
      return expect(']');
    }
  }
};

// the three typed values (fromJSON :157-164); each returns false on a syntax or type error (s.err says which)
inline bool parse_nodes(Scanner &s, OneGraph &g) {      // vector<int> (:157)
  g.label.clear();
  return s.array([&] { Scanner::Num v; if (!s.number(&v)) return false; g.label.push_back((uint32_t)v.as_int()); return true; });
}
inline bool parse_centers(Scanner &s, OneGraph &g) {    // vector<Vector3f> (:160, jsonToVector3f :147-153)
  g.xyz.clear();
  return s.array([&] {
    int k = 0;
    const bool in = s.array([&] { Scanner::Num v; if (!s.number(&v)) return false; if (k < 3) g.xyz.push_back(v.as_float()); k++; return true; });
    if (in && k < 3) return s.fail("a center needs three coordinates");
    return in;
  });
}
inline bool parse_poses(Scanner &s, OneGraph &g) {      // vector<float> (:161)
  g.n_pose = 0;
  return s.array([&] { Scanner::Num v; if (!s.number(&v)) return false; if (g.n_pose < 12) g.pose[g.n_pose] = v.as_float(); g.n_pose++; return true; });
}

inline bool finish_graph(OneGraph &g, bool have_nodes, bool have_centers, bool have_poses) {
  if (!have_nodes) { g.error = "missing key \"nodes\""; return false; }     // json.at()/operator[] would throw
  if (!have_centers) { g.error = "missing key \"centers\""; return false; }
  if (!have_poses) { g.error = "missing key \"poses\""; return false; }
  // Graph2CloudL indexes label[i] for every center (utility.hpp:653-658): fewer labels is UB there
  if (g.label.size() != g.xyz.size() / 3) { g.error = "nodes and centers differ in length"; return false; }
  for (int k = g.n_pose; k < 12; k++) g.pose[k] = 0.f;
  return true;
}

// The exact form for documents that repeat one of the three keys (or whose first occurrence cannot be read): walk the
// top-level object once with every value skipped (any JSON value is legal where the key's other occurrence will be
// used), remember where the deciding occurrence of each key starts, then read those three.
inline bool parse_graph_two_pass(const char *text, size_t len, OneGraph &g, bool last_wins) {
  Scanner s{text, text + len, {}};
  const char *at[3] = {nullptr, nullptr, nullptr};
  if (!s.expect('{')) { g.error = s.err; return false; }
  if (!s.peek('}')) {
    while (true) {
      std::string key;
      if (!s.string(&key) || !s.expect(':')) { g.error = s.err; return false; }
      s.ws();
      const int k = key == "nodes" ? 0 : key == "centers" ? 1 : key == "poses" ? 2 : -1;
      if (k >= 0 && (last_wins || !at[k])) at[k] = s.p;
      if (!s.skip_value()) { g.error = s.err; return false; }
      if (s.peek(',')) { s.p++; continue; }
      if (!s.expect('}')) { g.error = s.err; return false; }
      break;
    }
  } else {
    s.p++;
  }
  bool (*const read[3])(Scanner &, OneGraph &) = {parse_nodes, parse_centers, parse_poses};
  for (int k = 0; k < 3; k++) {
    if (!at[k]) continue;
    Scanner v{at[k], text + len, {}};
    if (!read[k](v, g)) { g.error = v.err; return false; }
  }
  return finish_graph(g, at[0] != nullptr, at[1] != nullptr, at[2] != nullptr);
}

// last_wins: which occurrence of a repeated key counts.  nlohmann::json 3.1.1 (the header this image carries, the one the
// differential test runs) emplaces — the FIRST stays; from 3.2 on (the SAX DOM parser; ROS noetic / Ubuntu 20.04 ship 3.7.3,
// the reference's CMake pins no version) object[key] is assigned — the LAST stays.  The producer (get_json.cpp:332-341) never
// writes a key twice.  sgtd_graphs_load takes the policy from SGTD_JSON_DUPLICATE_KEYS = first | last (default last).
inline bool parse_graph(const char *text, size_t len, OneGraph &g, bool last_wins = true) {
  Scanner s{text, text + len, {}};
  bool have_nodes = false, have_centers = false, have_poses = false, repeated = false;
  bool ok = s.expect('{');
  if (ok && s.peek('}')) { g.error = "missing key \"nodes\""; return false; }
  while (ok) {
    std::string key;
    if (!s.string(&key) || !s.expect(':')) { ok = false; break; }
    if (key == "nodes" && !have_nodes) { have_nodes = true; ok = parse_nodes(s, g); }
    else if (key == "centers" && !have_centers) { have_centers = true; ok = parse_centers(s, g); }
    else if (key == "poses" && !have_poses) { have_poses = true; ok = parse_poses(s, g); }
    else {                                      // another key, or a second occurrence of one of the three
      repeated = repeated || key == "nodes" || key == "centers" || key == "poses";
      ok = s.skip_value();
    }
    if (!ok) break;
    if (s.peek(',')) { s.p++; continue; }
    ok = s.expect('}');
    break;
  }
  // One pass is exact when every key occurred once and could be read.  Otherwise, if the last occurrence counts, the
  // two-pass form decides (a first occurrence of another type, or a repeated key: rare, and errors are not on the fast path).
  if (last_wins && (repeated || !ok)) {
    OneGraph h;
    const bool ok2 = parse_graph_two_pass(text, len, h, true);
    g.xyz = std::move(h.xyz); g.label = std::move(h.label); g.n_pose = h.n_pose; g.error = h.error;
    memcpy(g.pose, h.pose, sizeof(g.pose));
    return ok2;
  }
  if (!ok) { g.error = s.err; return false; }
  return finish_graph(g, have_nodes, have_centers, have_poses);
}

inline bool read_file(const std::string &path, std::string &out) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  out.resize(n > 0 ? (size_t)n : 0);
  const size_t got = n > 0 ? fread(&out[0], 1, (size_t)n, f) : 0;
  fclose(f);
  return got == out.size();
}

// frames in the order of `paths` (the caller decides: the reference's map order is the
// unsorted directory order, its query order is sorted — quirk 13)
inline bool load(const char *const *paths, int n_files, int n_threads, sgtd_graph_batch &b) {
  std::vector<OneGraph> graphs((size_t)n_files);
  std::atomic<int> next{0};
  const char *dup = getenv("SGTD_JSON_DUPLICATE_KEYS");
  const bool last_wins = !(dup && !strcmp(dup, "first"));
  auto work = [&] {
    std::string text;
    for (int i = next++; i < n_files; i = next++) {
      OneGraph g;   // parsed thread-locally: neighbouring slots of `graphs` share cache lines
      if (!read_file(paths[i], text)) g.error = std::string("Error opening file: ") + paths[i];   // Semantic_Graph.hpp:172-174
      else if (!parse_graph(text.data(), text.size(), g, last_wins)) g.error = std::string(paths[i]) + ": " + g.error;
      graphs[i] = std::move(g);
    }
  };
  const int nt = n_threads < 1 ? 1 : (n_threads > n_files ? (n_files > 0 ? n_files : 1) : n_threads);
  std::vector<std::thread> pool;
  for (int t = 1; t < nt; t++) pool.emplace_back(work);
  work();
  for (auto &t : pool) t.join();
  b.kp_off.assign(1, 0);
  b.xyz.clear(); b.label.clear(); b.poses.clear();
  size_t total = 0;
  for (auto &g : graphs) {
    if (!g.error.empty()) { b.error = g.error; return false; }
    total += g.label.size();
  }
  b.xyz.reserve(total * 3); b.label.reserve(total); b.poses.reserve((size_t)n_files * 12);
  for (auto &g : graphs) {
    b.xyz.insert(b.xyz.end(), g.xyz.begin(), g.xyz.end());
    b.label.insert(b.label.end(), g.label.begin(), g.label.end());
    b.poses.insert(b.poses.end(), g.pose, g.pose + 12);
    b.kp_off.push_back((int64_t)b.label.size());
  }
  return true;
}

// binary cache of a parsed batch: magic, counts, then the four arrays
static const char kMagic[8] = {'S', 'G', 'T', 'D', 'G', 'B', '0', '1'};

inline bool save_cache(const sgtd_graph_batch &b, const char *path) {
  FILE *f = fopen(path, "wb");
  if (!f) return false;
  const int64_t nf = (int64_t)b.kp_off.size() - 1, nk = (int64_t)b.label.size();
  bool ok = fwrite(kMagic, 1, 8, f) == 8 && fwrite(&nf, 8, 1, f) == 1 && fwrite(&nk, 8, 1, f) == 1;
  ok = ok && fwrite(b.kp_off.data(), 8, b.kp_off.size(), f) == b.kp_off.size();
  ok = ok && fwrite(b.poses.data(), 4, b.poses.size(), f) == b.poses.size();
  ok = ok && fwrite(b.label.data(), 4, b.label.size(), f) == b.label.size();
  ok = ok && fwrite(b.xyz.data(), 4, b.xyz.size(), f) == b.xyz.size();
  return fclose(f) == 0 && ok;
}

inline bool load_cache(const char *path, sgtd_graph_batch &b) {
  FILE *f = fopen(path, "rb");
  if (!f) { b.error = std::string("Error opening file: ") + path; return false; }
  char magic[8];
  int64_t nf = 0, nk = 0;
  bool ok = fread(magic, 1, 8, f) == 8 && !memcmp(magic, kMagic, 8) && fread(&nf, 8, 1, f) == 1 && fread(&nk, 8, 1, f) == 1 &&
            nf >= 0 && nk >= 0;
  if (ok) {
    // the header's counts must agree with the file's size before anything is allocated from them
    ok = fseek(f, 0, SEEK_END) == 0;
    const long long fsize = ok ? (long long)ftell(f) : -1;
    ok = ok && nf < (1ll << 40) && nk < (1ll << 40) && fsize == 24 + (nf + 1) * 8 + nf * 48 + nk * 16 &&
         fseek(f, 24, SEEK_SET) == 0;
  }
  if (ok) {
    b.kp_off.resize((size_t)nf + 1); b.poses.resize((size_t)nf * 12); b.label.resize((size_t)nk); b.xyz.resize((size_t)nk * 3);
    ok = fread(b.kp_off.data(), 8, b.kp_off.size(), f) == b.kp_off.size() && fread(b.poses.data(), 4, b.poses.size(), f) == b.poses.size() &&
         fread(b.label.data(), 4, b.label.size(), f) == b.label.size() && fread(b.xyz.data(), 4, b.xyz.size(), f) == b.xyz.size() &&
         b.kp_off.front() == 0 && b.kp_off.back() == nk;
    for (size_t k = 1; ok && k < b.kp_off.size(); k++) ok = b.kp_off[k] >= b.kp_off[k - 1];   // frames are ranges
  }
  fclose(f);
  if (!ok) {
    b.kp_off.assign(1, 0); b.poses.clear(); b.label.clear(); b.xyz.clear();
    b.error = std::string(path) + ": not a graph-batch cache";
  }
  return ok;
}

}  // namespace ingest
