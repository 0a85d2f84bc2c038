// Micro-benchmark: issue rate of the VALU instructions a 255-bit modular multiply can be built from.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu
// Reports wave64-instruction issue cost in cycles per SIMD (relative to wall clock at the measured rate).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 8;

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3u + blockIdx.x;
    uint64_t acc[UNROLL];
    double   dacc[UNROLL];
    uint32_t w[UNROLL];
    for (int i = 0; i < UNROLL; i++) { acc[i] = a + i; dacc[i] = (double)(a + i); w[i] = b + i; }
    double da = (double)a * 1.000001, db = (double)b * 0.999;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            if constexpr (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(w[i]) : "vcc");
            if constexpr (OP == 1) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(w[i]) : "v"(a));
            if constexpr (OP == 2) asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(w[i]) : "v"(a));
            if constexpr (OP == 3) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(w[i]) : "v"(a));
            if constexpr (OP == 4) asm volatile("v_mul_hi_u32_u24 %0, %1, %0" : "+v"(w[i]) : "v"(a));
            if constexpr (OP == 5) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(w[i]) : "v"(a), "v"(b));
            if constexpr (OP == 6) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(dacc[i]) : "v"(da), "v"(db));
            if constexpr (OP == 7) asm volatile("v_add_f64 %0, %1, %0" : "+v"(dacc[i]) : "v"(da));
            if constexpr (OP == 8) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(w[i]) : "v"(a), "v"(b));
            if constexpr (OP == 9) asm volatile("v_add_co_u32 %0, vcc, %1, %0" : "+v"(w[i]) : "v"(a) : "vcc");
            if constexpr (OP == 10) asm volatile("v_addc_co_u32 %0, vcc, %1, %0, vcc" : "+v"(w[i]) : "v"(a) : "vcc");
            if constexpr (OP == 11) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(w[i]) : "v"(a), "v"(b));
            if constexpr (OP == 12) asm volatile("v_lshl_add_u32 %0, %1, 3, %0" : "+v"(w[i]) : "v"(a));
            if constexpr (OP == 13) asm volatile("v_alignbit_b32 %0, %1, %0, 7" : "+v"(w[i]) : "v"(a));
            if constexpr (OP == 14) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(w[i]) : "v"(a), "v"(b));
            if constexpr (OP == 15) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(w[i]) : "v"(a), "v"(b));
            if constexpr (OP == 16) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_add_co_u32 %1, vcc, %2, %1" : "+v"(acc[i]), "+v"(w[i]) : "v"(a), "v"(b) : "vcc");
            if constexpr (OP == 17) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(dacc[i]) : "v"(da));
            if constexpr (OP == 18) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(w[i]) : "v"(a));
            if constexpr (OP == 19) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(acc[i]));
            if constexpr (OP == 20) asm volatile("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(w[i]) : "v"(a), "v"(b));
            if constexpr (OP == 21) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(acc[(i+1)%UNROLL]), "v"(acc[(i+2)%UNROLL]));
            if constexpr (OP == 22) asm volatile("v_mul_lo_u32 %0, %2, %0\n\tv_fma_f64 %1, %3, %4, %1" : "+v"(w[i]), "+v"(dacc[i]) : "v"(a), "v"(da), "v"(db));
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < UNROLL; i++) r ^= (uint32_t)acc[i] ^ (uint32_t)(acc[i] >> 32) ^ w[i] ^ (uint32_t)dacc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP> int run(const char* name, int nper, uint32_t* d_out, int blocks) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 7u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 7u + r);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    // wave-instructions per SIMD: blocks*4 waves / (256 CUs*4 SIMDs) * ITERS*UNROLL*nper
    double waves_per_simd = (double)blocks * 4 / (256.0 * 4);
    double winst = waves_per_simd * ITERS * UNROLL * nper;
    double ns_per = ms * 1e6 / winst;
    printf("%-28s %8.3f ms  %7.3f ns/wave-inst/SIMD  = %6.2f cyc @2.4GHz  %6.2f cyc @2.0GHz\n", name, ms, ns_per, ns_per * 2.4, ns_per * 2.0);
    return 0;
}

int main() {
    int blocks = 256 * 8;  // 8 blocks of 4 waves per CU = 8 waves/SIMD
    uint32_t* d_out; CHECK(hipMalloc(&d_out, (size_t)blocks * 256 * 4));
    run<8>("v_fma_f32", 1, d_out, blocks);
    run<0>("v_mad_u64_u32", 1, d_out, blocks);
    run<1>("v_mul_lo_u32", 1, d_out, blocks);
    run<2>("v_mul_hi_u32", 1, d_out, blocks);
    run<3>("v_mul_u32_u24", 1, d_out, blocks);
    run<4>("v_mul_hi_u32_u24", 1, d_out, blocks);
    run<5>("v_mad_u32_u24", 1, d_out, blocks);
    run<20>("v_mad_u32_u16", 1, d_out, blocks);
    run<6>("v_fma_f64", 1, d_out, blocks);
    run<7>("v_add_f64", 1, d_out, blocks);
    run<17>("v_mul_f64", 1, d_out, blocks);
    run<9>("v_add_co_u32", 1, d_out, blocks);
    run<10>("v_addc_co_u32", 1, d_out, blocks);
    run<11>("v_add3_u32", 1, d_out, blocks);
    run<12>("v_lshl_add_u32", 1, d_out, blocks);
    run<13>("v_alignbit_b32", 1, d_out, blocks);
    run<18>("v_xor_b32", 1, d_out, blocks);
    run<19>("v_lshlrev_b64", 1, d_out, blocks);
    run<14>("v_dot4_u32_u8", 1, d_out, blocks);
    run<15>("v_dot2_u32_u16", 1, d_out, blocks);
    run<21>("v_pk_fma_f32", 1, d_out, blocks);
    run<16>("mad_u64_u32 + add_co (pair)", 2, d_out, blocks);
    run<22>("mul_lo_u32 + fma_f64 (pair)", 2, d_out, blocks);
    return 0;
}
