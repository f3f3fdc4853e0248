// Host check of common.h's division-by-multiply-high (fast_div_setup + the device formula restated with a 64-bit product):
// exact for every dividend below 2^31 that matters -- all small divisors exhaustively around multiples, the extremes, and
// random dividends for large divisors.  Built and run by tests/test_cpu_host.py with hipcc (host code only).
#include "common.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>

static inline int fast_div_host(int n, unsigned mul, unsigned shr) {
  return mul ? (int)((unsigned)(((uint64_t)(unsigned)n * mul) >> 32) >> shr) : n;
}

int main() {
  long bad = 0, checked = 0;
  auto check = [&](unsigned d, int64_t n, unsigned mul, unsigned shr) {
    if (n < 0 || n > 0x7fffffff) return;
    ++checked;
    if (fast_div_host((int)n, mul, shr) != (int)(n / d) && bad++ < 5) printf("bad: %lld / %u\n", (long long)n, d);
  };
  for (unsigned d = 1; d <= 70000; ++d) {
    unsigned mul, shr;
    d3f::fast_div_setup(d, &mul, &shr);
    for (int64_t k = 0; k < 40; ++k) {
      const int64_t base = (k < 20) ? k * d : (0x7fffffffll / d - (k - 20)) * d;
      for (int e = -2; e <= 2; ++e) check(d, base + e, mul, shr);
    }
  }
  srand(7);
  const unsigned big[] = {1u << 16, 1u << 20, 65536u * 48u, 448u * 448u, 1000003u, 0x40000000u, 0x7fffffffu};
  for (unsigned d : big) {
    unsigned mul, shr;
    d3f::fast_div_setup(d, &mul, &shr);
    for (int r = 0; r < 200000; ++r) check(d, (int64_t)((((uint64_t)rand() << 16) ^ (uint64_t)rand()) & 0x7fffffff), mul, shr);
    for (int e = -3; e <= 0; ++e) check(d, 0x7fffffffll + e, mul, shr);
  }
  printf("checked %ld bad %ld\n", checked, bad);
  return bad != 0;
}
