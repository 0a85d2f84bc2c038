// Device-vs-host check of the scalar arithmetic in dapol_amd/csrc/sc.h (same header, both sides of one hipcc build).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I dapol_amd/csrc tools/sc_selftest.hip -o build/sc_selftest
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "sc.h"
using namespace dapol;

__host__ __device__ inline void ops(const uint32_t* in, uint32_t* out) {
    sc a, b, r;
    for (int i = 0; i < 8; i++) { a.v[i] = in[i]; b.v[i] = in[8 + i]; }
    sc_montmul(r, a, b);
    for (int i = 0; i < 8; i++) out[i] = r.v[i];
    sc ar;                          // reduce a below L first: a * R^2 / R = aR mod L, then back
    { sc k; for (int i = 0; i < 8; i++) k.v[i] = SC_R2[i]; sc_montmul(ar, a, k); }
    sc_add(r, ar, b);
    for (int i = 0; i < 8; i++) out[8 + i] = r.v[i];
    sc_sub(r, ar, b);
    for (int i = 0; i < 8; i++) out[16 + i] = r.v[i];
    sc_sub(r, b, ar);
    for (int i = 0; i < 8; i++) out[24 + i] = r.v[i];
}
__global__ void k(const uint32_t* in, uint32_t* out, int n) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) ops(in + 16 * t, out + 32 * t);
}
int main() {
    const int n = 1 << 16;
    std::vector<uint32_t> in(16 * n), out(32 * n), ref(32 * n);
    uint64_t s = 0x9e3779b97f4a7c15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
    for (int t = 0; t < n; t++) {
        for (int i = 0; i < 16; i++) in[16 * t + i] = (t % 7 == 0) ? 0xffffffffu : rnd();
        in[16 * t + 15] &= 0x0fffffffu;                       // b < 2^252 < L
        if (t % 5 == 0) for (int i = 8; i < 16; i++) in[16 * t + i] = SC_L[i - 8] - (i == 8 ? 1 + (t & 3) : 0);   // b = L - 1 - small
    }
    uint32_t *d_in, *d_out;
    hipMalloc(&d_in, in.size() * 4); hipMalloc(&d_out, out.size() * 4);
    hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    #ifndef BLK
#define BLK 64
#endif
    hipLaunchKernelGGL(k, dim3(n / BLK), dim3(BLK), 0, 0, d_in, d_out, n);
    hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost);
    int bad[4] = {0, 0, 0, 0};
    for (int t = 0; t < n; t++) {
        ops(&in[16 * t], &ref[32 * t]);
        for (int o = 0; o < 4; o++)
            for (int i = 0; i < 8; i++)
                if (out[32 * t + 8 * o + i] != ref[32 * t + 8 * o + i]) { bad[o]++; break; }
    }
    printf("mismatches: montmul %d, add %d, sub %d, sub(rev) %d of %d\n", bad[0], bad[1], bad[2], bad[3], n);
    for (int t = 0; t < n && (bad[0] + bad[1] + bad[2] + bad[3]); t++) {
        bool b0 = false;
        for (int i = 0; i < 32; i++) b0 |= out[32 * t + i] != ref[32 * t + i];
        if (b0) {
            printf("first bad t=%d\n in a:", t); for (int i = 0; i < 8; i++) printf(" %08x", in[16 * t + i]);
            printf("\n in b:"); for (int i = 8; i < 16; i++) printf(" %08x", in[16 * t + i]);
            for (int o = 0; o < 4; o++) { printf("\n dev%d:", o); for (int i = 0; i < 8; i++) printf(" %08x", out[32 * t + 8 * o + i]); printf("\n ref%d:", o); for (int i = 0; i < 8; i++) printf(" %08x", ref[32 * t + 8 * o + i]); }
            printf("\n");
            break;
        }
    }
    return 0;
}
