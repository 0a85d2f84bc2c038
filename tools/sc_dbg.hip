#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "sc.h"
using namespace dapol;
__host__ __device__ inline void dbg(const uint32_t* in, uint64_t* out) {
    sc a, b;
    for (int i = 0; i < 8; i++) { a.v[i] = in[i]; b.v[i] = in[8 + i]; }
    const uint32_t M29 = 0x1fffffffu;
    const uint32_t L0 = 0x1cf5d3edu, L1 = 0x009318d2u, L2 = 0x1de73596u, L3 = 0x1df3bd45u, L4 = 0x0000014du, L8 = 0x00100000u;
    uint32_t A[9], B[9];
    for (int k = 0; k < 9; k++) {
        const int o = 29 * k, w = o >> 5, sh = o & 31;
        uint32_t xa = a.v[w] >> sh, xb = b.v[w] >> sh;
        if (sh > 3 && w + 1 < 8) { xa |= a.v[w + 1] << (32 - sh); xb |= b.v[w + 1] << (32 - sh); }
        A[k] = xa & M29;
        B[k] = xb & M29;
        out[k] = A[k]; out[9 + k] = B[k];
    }
    uint64_t c[18];
    for (int k = 0; k < 18; k++) c[k] = 0;
    for (int i = 0; i < 9; i++)
        for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)A[i] * B[j];
    for (int k = 0; k < 18; k++) out[18 + k] = c[k];
    for (int i = 0; i < 9; i++) {
        const uint32_t mask = i < 8 ? M29 : 0x00ffffffu;
        const uint32_t m = ((uint32_t)c[i] * SC_LFACTOR) & mask;
        out[54 + i] = m;
        c[i] += (uint64_t)m * L0;
        c[i + 1] += (uint64_t)m * L1;
        c[i + 2] += (uint64_t)m * L2;
        c[i + 3] += (uint64_t)m * L3;
        c[i + 4] += (uint64_t)m * L4;
        c[i + 8] += (uint64_t)m * L8;
        if (i < 8) c[i + 1] += c[i] >> 29;
    }
    for (int k = 0; k < 18; k++) out[36 + k] = c[k];
    uint32_t lim[11];
    for (int k = 8; k < 17; k++) {
        lim[k - 8] = (uint32_t)c[k] & M29;
        c[k + 1] += c[k] >> 29;
    }
    lim[9] = (uint32_t)c[17];
    lim[10] = 0;
    for (int k = 0; k < 11; k++) out[64 + k] = lim[k];
    uint32_t t[8];
    for (int j = 0; j < 8; j++) {
        const int o = 24 + 32 * j, q = o / 29, sh = o % 29;
        uint32_t x = lim[q] >> sh;
        x |= lim[q + 1] << (29 - sh);
        if (58 - sh < 32) x |= lim[q + 2] << (58 - sh);
        t[j] = x;
        out[80 + j] = x;
    }
    sc r;
    sc_cond_sub(r, t, 0);
    for (int j = 0; j < 8; j++) out[90 + j] = r.v[j];
    sc r2;
    sc_montmul(r2, a, b);
    for (int j = 0; j < 8; j++) out[100 + j] = r2.v[j];
}
__global__ void k(const uint32_t* in, uint64_t* out) { dbg(in, out); }
int main() {
    uint32_t in[16] = {0xffffffff,0xffffffff,0xffffffff,0xffffffff,0xffffffff,0xffffffff,0xffffffff,0xffffffff, 0x5cf5d3ec,0x5812631a,0xa2f79cd6,0x14def9de,0,0,0,0x10000000};
    uint64_t out[128], ref[128];
    uint32_t* d_in; uint64_t* d_out;
    hipMalloc(&d_in, 64); hipMalloc(&d_out, 1024);
    hipMemcpy(d_in, in, 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d_in, d_out);
    hipMemcpy(out, d_out, 1024, hipMemcpyDeviceToHost);
    dbg(in, ref);
    const char* names[] = {"A", "B", "prod", "prod", "red", "red", "m"};
    for (int i = 0; i < 108; i++) if (out[i] != ref[i]) printf("diff at %d: dev %016llx ref %016llx\n", i, (unsigned long long)out[i], (unsigned long long)ref[i]);
    printf("LFACTOR host %08x\n", SC_LFACTOR);
    return 0;
}
