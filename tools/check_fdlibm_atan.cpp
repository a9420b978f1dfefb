// Exhaustive pin of the oracle's fdlibm restatement (oracle/lo_math.h: fd_atanf / fd_atan2f) against THIS image's libm:
//   g++ -O2 -ffp-contract=off -fno-builtin -I oracle tools/check_fdlibm_atan.cpp -o /tmp/check_fdlibm && /tmp/check_fdlibm
// all 2^32 atanf arguments, 3e8 atan2f pairs (half lidar-like coordinates, half random bit patterns); ~45 s on one core.
#include "lo_math.h"
using lo::fbits; using lo::bitsf; using lo::fd_atanf; using lo::fd_atan2f;
#include <cstdio>
#include <random>
int main() {
    // atanf: EVERY float
    unsigned long long bad = 0, n = 0;
    for (uint64_t u = 0; u < (1ull << 32); u += 1) {
        float x = bitsf((uint32_t)u);
        float a = atanf(x), b = fd_atanf(x);
        if (fbits(a) != fbits(b) && !(a != a && b != b)) { if (bad < 5) printf("atanf(%a): libm %a fd %a\n", x, a, b); bad++; }
        n++;
    }
    printf("atanf: %llu inputs, %llu mismatches\n", n, bad);
    // atan2f: random pairs in lidar-like ranges + random bit patterns
    std::mt19937_64 rng(1);
    bad = 0; n = 0;
    std::uniform_real_distribution<float> U(-200.f, 200.f);
    for (long i = 0; i < 300000000L; i++) {
        float y, x;
        if (i & 1) { y = U(rng); x = U(rng); } else { uint64_t r = rng(); y = bitsf((uint32_t)r); x = bitsf((uint32_t)(r >> 32)); }
        float a = atan2f(y, x), b = fd_atan2f(y, x);
        if (fbits(a) != fbits(b) && !(a != a && b != b)) { if (bad < 5) printf("atan2f(%a,%a): libm %a fd %a\n", y, x, a, b); bad++; }
        n++;
    }
    printf("atan2f: %llu inputs, %llu mismatches\n", n, bad);
    return 0;
}
