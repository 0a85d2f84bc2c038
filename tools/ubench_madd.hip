// Micro-benchmark: the VALU-only cost of the point operations k_rp_msm is made of -- the same ge_madd / ge_dbl
// (dapol_amd/csrc/ge.h) in a register-resident loop, no table gathers, three wavefronts per SIMD like the kernel.
// The result is the instruction-issue roof of the MSM: launch time >= waves * (adds * t_add + dbls * t_dbl) / SIMDs.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I dapol_amd/csrc tools/ubench_madd.hip -o tools/ubench_madd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "ge.h"

using namespace dapol;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// OP 0: mixed additions (the table entry lives in registers; its sign flips with the iteration so nothing is hoisted)
// OP 1: doublings without T, OP 2: doublings with T
template <int OP>
__global__ __launch_bounds__(64, 3) void k(int32_t* out, const int32_t* in, int iters, unsigned long long* clk) {
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    ge_p3 acc;
    ge_niels q;
    const int32_t* p = in + threadIdx.x * 70;
    for (int i = 0; i < FE_NL; i++) {
        acc.X.v[i] = p[i]; acc.Y.v[i] = p[10 + i]; acc.Z.v[i] = p[20 + i]; acc.T.v[i] = p[30 + i];
        q.ypx.v[i] = p[40 + i]; q.ymx.v[i] = p[50 + i]; q.xy2d.v[i] = p[60 + i];
    }
    for (int it = 0; it < iters; it++) {
        if constexpr (OP == 0) ge_madd(acc, acc, q, (it ^ threadIdx.x) & 1);
        if constexpr (OP == 1) ge_dbl(acc, acc, false);
        if constexpr (OP == 2) ge_dbl(acc, acc, true);
    }
    if (blockIdx.x == 100 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }   // shader clocks vs 100 MHz ticks
    int32_t* o = out + ((size_t)blockIdx.x * 64 + threadIdx.x) * 40;
    for (int i = 0; i < FE_NL; i++) { o[i] = acc.X.v[i]; o[10 + i] = acc.Y.v[i]; o[20 + i] = acc.Z.v[i]; o[30 + i] = acc.T.v[i]; }
}

template <int OP> int run(const char* name, int32_t* d_out, const int32_t* d_in, int blocks, int iters) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    static unsigned long long* d_clk = nullptr;
    if (!d_clk) CHECK(hipMalloc(&d_clk, 16));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d_out, d_in, iters, d_clk);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d_out, d_in, iters, d_clk);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    double waves_per_simd = (double)blocks / (256.0 * 4);
    double ns_per = ms * 1e6 / (waves_per_simd * iters);        // SIMD time per wave-operation
    unsigned long long h_clk[2];
    CHECK(hipMemcpy(h_clk, d_clk, 16, hipMemcpyDeviceToHost));
    double ghz = (double)h_clk[0] / ((double)h_clk[1] * 10.0);          // wall_clock64 ticks at 100 MHz
    printf("%-24s %8.3f ms  %9.1f ns of SIMD time per wave-op = %7.1f cycles at the measured %.3f GHz shader clock (grid of %d waves)\n", name, ms,
           ns_per, ns_per * ghz, ghz, blocks);
    return 0;
}

int main() {
    const int blocks = 3072 * 2;          // two rounds of 3 resident wavefronts per SIMD
    int32_t h_in[64 * 70];
    // the basepoint in extended form and its niels form would do; any reduced limbs exercise the same instructions
    for (int t = 0; t < 64; t++)
        for (int i = 0; i < 70; i++) h_in[t * 70 + i] = (int32_t)(((uint32_t)(t * 2654435761u + i * 40503u + 12345u)) & ((i % 10) == 8 ? 0x7fffff : 0x1fffffff));     // reduced 29-bit limbs (23 bits in limb 8)
    int32_t *d_in, *d_out;
    CHECK(hipMalloc(&d_in, sizeof(h_in))); CHECK(hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_out, (size_t)blocks * 64 * 40 * 4));
    for (int rep = 0; rep < 2; rep++) {
        run<0>("ge_madd (mixed add)", d_out, d_in, blocks, 4096);
        run<1>("ge_dbl (no T)", d_out, d_in, blocks, 4096);
        run<2>("ge_dbl (with T)", d_out, d_in, blocks, 4096);
    }
    // fewer resident wavefronts: one round of w waves per SIMD (does one wave alone keep the VALU issuing?)
    for (int w = 1; w <= 3; w++) {
        printf("-- %d wave(s) per SIMD: ", w);
        run<0>("ge_madd", d_out, d_in, 1024 * w, 4096);
    }
    return 0;
}
