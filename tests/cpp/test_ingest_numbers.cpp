// The ingest's number scanner (sgtd_amd/csrc/graph_ingest.hip.h, Scanner::number) against the C library on random
// tokens: a token with a fraction or an exponent must give the double strtod gives — the scanner takes a shortcut
// through x87 extended precision for tokens of up to 19 significant digits (fast_double) — and an integer token the
// value strtoull / strtoll give, with nlohmann::json 3.1.1's three number kinds.
//   g++ -std=c++17 -O2 -x c++ tests/cpp/test_ingest_numbers.cpp -o t && ./t [tokens]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>

#include "../../sgtd_amd/csrc/graph_ingest.hip.h"

int main(int argc, char **argv) {
  const long long n = argc > 1 ? atoll(argv[1]) : 5000000;
  std::mt19937_64 rng(20251121);
  long long floats = 0, ints = 0, bad = 0;
  char buf[64];
  for (long long it = 0; it < n; it++) {
    std::string tok;
    switch (rng() % 6) {
      case 0: {   // a float32 value the way the producer's library dumps it (17 significant digits)
        float f; const uint32_t b = (uint32_t)rng(); memcpy(&f, &b, 4);
        if (!(f - f == 0)) continue;
        snprintf(buf, sizeof buf, "%.17g", (double)f); tok = buf; break;
      }
      case 1: {   // metre-sized coordinates, 9 .. 17 digits
        const float f = (float)((double)((long long)(rng() % 2000001) - 1000000) / 1000.0 + (double)(rng() % 1000) / 1e6);
        snprintf(buf, sizeof buf, "%.*g", 9 + (int)(rng() % 9), (double)f); tok = buf; break;
      }
      case 2: {   // random digit strings with a point somewhere, sometimes an exponent
        const int nd = 1 + (int)(rng() % 21);
        if (rng() & 1) tok += "-";
        const int dot = (int)(rng() % (nd + 1));
        for (int i = 0; i < nd; i++) {
          if (i == dot && i) tok += ".";
          tok += (char)('0' + (i == 0 && nd > 1 && dot != 1 ? 1 + rng() % 9 : rng() % 10));
        }
        if (tok.back() == '.') tok += "0";
        if (rng() % 3 == 0) {
          const int ex = (int)(rng() % 61) - 30;
          snprintf(buf, sizeof buf, "%c%s%d", rng() & 1 ? 'e' : 'E', ex >= 0 && (rng() & 1) ? "+" : "", ex);
          tok += buf;
        }
        break;
      }
      case 3: {   // any finite double, 1 .. 17 digits
        double d; const uint64_t b = rng(); memcpy(&d, &b, 8);
        if (!(d - d == 0)) continue;
        snprintf(buf, sizeof buf, "%.*g", 1 + (int)(rng() % 17), d); tok = buf; break;
      }
      case 4: {   // integers around the limits of the three kinds
        static const char *edge[] = {"0", "-0", "9223372036854775807", "9223372036854775808", "-9223372036854775808", "-9223372036854775809",
                                     "18446744073709551615", "18446744073709551616", "999999999999999999", "-999999999999999999",
                                     "1000000000000000000", "-1000000000000000000", "9999999999999999999", "10000000000000000000"};
        tok = edge[rng() % (sizeof edge / sizeof edge[0])]; break;
      }
      default: {  // plain integers of 1 .. 20 digits
        const int nd = 1 + (int)(rng() % 20);
        if (rng() & 1) tok += "-";
        for (int i = 0; i < nd; i++) tok += (char)('0' + (i == 0 && nd > 1 ? 1 + rng() % 9 : rng() % 10));
      }
    }
    if (tok.find("inf") != std::string::npos || tok.find("nan") != std::string::npos) continue;
    // (a leading zero before more digits is not JSON: "007" — the generators above do not make one except case 2's "0x" forms)
    if (tok.size() > 1 && tok[tok[0] == '-'] == '0' && tok.size() > (size_t)(tok[0] == '-') + 1 && tok[(tok[0] == '-') + 1] >= '0' && tok[(tok[0] == '-') + 1] <= '9') continue;
    ingest::Scanner s{tok.data(), tok.data() + tok.size(), {}};
    ingest::Scanner::Num v;
    if (!s.number(&v) || s.p != s.end) { if (bad++ < 10) printf("NOT PARSED %s (%s)\n", tok.c_str(), s.err.c_str()); continue; }
    const bool is_float = tok.find_first_of(".eE") != std::string::npos;
    bool same;
    if (is_float) {
      const double ref = strtod(tok.c_str(), nullptr);
      same = v.kind == 2 && !memcmp(&v.d, &ref, 8);
      floats++;
    } else {
      errno = 0;
      if (tok[0] == '-') {
        const long long ref = strtoll(tok.c_str(), nullptr, 10);
        const double dref = strtod(tok.c_str(), nullptr);
        same = errno == 0 ? (v.kind == 1 && v.i == ref) : (v.kind == 2 && !memcmp(&v.d, &dref, 8));
      } else {
        const unsigned long long ref = strtoull(tok.c_str(), nullptr, 10);
        const double dref = strtod(tok.c_str(), nullptr);
        same = errno == 0 ? (v.kind == 0 && v.u == ref) : (v.kind == 2 && !memcmp(&v.d, &dref, 8));
      }
      ints++;
    }
    if (!same && bad++ < 10) printf("MISMATCH %s: kind %d u %llu i %lld d %.17g\n", tok.c_str(), v.kind, (unsigned long long)v.u, (long long)v.i, v.d);
  }
  printf("%lld float tokens, %lld integer tokens, %lld differences\n", floats, ints, bad);
  if (!bad) printf("number scanner equals the C library\n");
  return bad != 0;
}
