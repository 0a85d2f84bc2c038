// Host check of sc_invert_vartime_mont (dapol_amd/csrc/sc.h, the same header the kernels compile) against the Fermat ladder
// sc_invert_mont and against a * a^-1 = 1: edge values (0, 1, L - 1, small, near L, single bits) and random scalars.
// usage: sc_invert_vartime_test [count]
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include "sc.h"
using namespace dapol;
int main(int argc, char** argv) {
    uint64_t s = 0x9e3779b97f4a7c15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
    int bad = 0;
    const int n = argc > 1 ? atoi(argv[1]) : 200000;
    for (int t = 0; t < n; t++) {
        sc a, r1, r2, k;
        for (int i = 0; i < 8; i++) a.v[i] = rnd();
        a.v[7] &= 0x0fffffffu;
        if (t == 0) sc_zero(a);
        if (t == 1) { sc_zero(a); a.v[0] = 1; }
        if (t == 2) { for (int i = 0; i < 8; i++) a.v[i] = SC_L[i]; a.v[0] -= 1; }
        if (t == 3) sc_one_mont(a);
        if (t >= 4 && t < 300) { sc_zero(a); a.v[0] = t; }
        if (t >= 300 && t < 600) { for (int i = 0; i < 8; i++) a.v[i] = SC_L[i]; a.v[0] -= (t - 298); }
        if (t >= 600 && t < 900) { sc_zero(a); a.v[(t / 32) % 8] = 1u << (t % 32); a.v[7] &= 0x0fffffffu; }
        sc_invert_mont(r1, a);
        sc_invert_vartime_mont(r2, a);
        if (memcmp(r1.v, r2.v, 32)) { if (bad < 5) printf("mismatch at %d\n", t); bad++; }
        if (t > 0) { sc_montmul(k, r2, a); sc one; sc_one_mont(one); if (memcmp(k.v, one.v, 32) && !sc_is_zero(a)) { if (bad < 5) printf("not inverse at %d\n", t); bad++; } }
    }
    printf("%d tested, %d bad\n", n, bad);
    return bad != 0;
}
