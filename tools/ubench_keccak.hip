// Micro-benchmark: Keccak-f[1600] held by ONE WAVEFRONT (one 64-bit state word per lane, lane = x + 5 y), the serial floor of the
// batched verifier's transcript replay (253 dependent permutations per 1,024-party proof, one wavefront per proof, one wavefront
// per SIMD: nothing hides its latency).  Variants of the same permutation, checked against the one-lane reference of hash.h:
//   0  the permutation as the library ran it until round 5 (__shfl, 64-bit shifts for rho, the round constant loaded in a lane-0
//      branch at the end of the round)
//   1  the same stages through ds_bpermute with byte addresses computed once, rounds unrolled (round constants as literals),
//      rho by two v_alignbit and a per-lane swap
//   2  the stages through LDS: 64-bit ds_write2 / ds_read2 on doubled rows (no modular index), 8 LDS instructions per round
//      instead of 18 ds_bpermute
//   3-5  as 1 with the rounds in a loop (1 / 2 / 4 per trip): the round constant loaded at the top of the round, applied by mask
//   6  keccak_f1600_wave of hash.h as the library runs it now (= 4)
// Tried and dropped: rows of eight lanes with the column parities by DPP row_ror:8 + v_permlane32_swap / v_permlane16_swap and
// theta's / chi's neighbours by DPP row shifts, leaving one ds_bpermute stage (rho/pi) per round -- 544 cycles per round against
// 491: a lone wavefront pays ~8 cycles per DEPENDENT VALU instruction, so ~25 more of them cost what two crossbar trips did.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I dapol_amd/csrc tools/ubench_keccak.hip -o build/ubench_keccak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include "hash.h"

using namespace dapol;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __constant__ const uint64_t KRC[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull, 0x000000000000808Bull, 0x0000000080000001ull,
    0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008Aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000Aull,
    0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull, 0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull,
    0x000000000000800Aull, 0x800000008000000Aull, 0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};

// The permutation as the library ran it until round 5 (kept here as the baseline of the comparison).
__device__ __forceinline__ uint64_t shfl64_(uint64_t v, int src) {
    uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src, 64), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src, 64);
    return ((uint64_t)hi << 32) | lo;
}
struct OldLanes { int th1, th2, th3, th4, xp1, xp2, pi_src, cm_src, cp_src, rot_src; };
__device__ __forceinline__ void old_lanes_init(OldLanes& K, int l) {
    const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    if (l >= 25) { K.th1 = K.th2 = K.th3 = K.th4 = K.xp1 = K.xp2 = K.pi_src = K.cm_src = K.cp_src = l; K.rot_src = 0; return; }
    int x = l % 5, y = l / 5, row = 5 * y;
    K.th1 = (l + 5) % 25; K.th2 = (l + 10) % 25; K.th3 = (l + 15) % 25; K.th4 = (l + 20) % 25;
    K.xp1 = row + (x + 1) % 5; K.xp2 = row + (x + 2) % 5;
    int sy = x, sx = (3 * ((y - 3 * x) % 5 + 5)) % 5;
    K.pi_src = sx + 5 * sy;
    K.cm_src = (sx + 4) % 5;
    K.cp_src = (sx + 1) % 5;
    int r = 0;
    for (int i = 0; i < 25; i++) r = (i == K.pi_src) ? ROT[i] : r;
    K.rot_src = r;
}
__device__ __forceinline__ uint64_t old_keccak_wave(uint64_t a, const OldLanes& K, int l) {
    for (int r = 0; r < 24; r++) {
        uint64_t as = shfl64_(a, K.pi_src);
        uint64_t c = a ^ shfl64_(a, K.th1) ^ shfl64_(a, K.th2) ^ shfl64_(a, K.th3) ^ shfl64_(a, K.th4);
        uint64_t cm = shfl64_(c, K.cm_src), cp = shfl64_(c, K.cp_src);
        as ^= cm ^ ((cp << 1) | (cp >> 63));
        uint64_t b = K.rot_src ? ((as << K.rot_src) | (as >> (64 - K.rot_src))) : as;
        uint64_t b1 = shfl64_(b, K.xp1), b2 = shfl64_(b, K.xp2);
        a = b ^ (~b1 & b2);
        if (l == 0) a ^= KRC[r];
    }
    return a;
}

struct KL2 {                          // byte addresses for ds_bpermute, rho as (swap halves?, shift < 32)
    int th1, th2, th3, th4, xp1, xp2, pi_src, cm_src, cp_src;
    uint32_t sh;                      // rotation amount mod 32
    bool swap;                        // rotation amount >= 32
};
__device__ __forceinline__ void kl2_init(KL2& K, int l) {
    OldLanes k;
    old_lanes_init(k, l);
    K.th1 = k.th1 << 2; K.th2 = k.th2 << 2; K.th3 = k.th3 << 2; K.th4 = k.th4 << 2; K.xp1 = k.xp1 << 2; K.xp2 = k.xp2 << 2;
    K.pi_src = k.pi_src << 2; K.cm_src = k.cm_src << 2; K.cp_src = k.cp_src << 2;
    K.sh = (uint32_t)k.rot_src & 31u; K.swap = k.rot_src >= 32;
}
struct W2 { uint32_t lo, hi; };
__device__ __forceinline__ W2 bperm(int addr, W2 v) {
    W2 r;
    r.lo = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)v.lo);
    r.hi = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)v.hi);
    return r;
}
__device__ __forceinline__ W2 x2(W2 a, W2 b) { return W2{a.lo ^ b.lo, a.hi ^ b.hi}; }
__device__ __forceinline__ W2 rotl_var(W2 a, uint32_t sh, bool swap) {        // rotl64 by sh + 32 swap, sh in [0, 32)
    // v_alignbit(hi, lo, 32 - sh) = (hi:lo >> (32 - sh)) low word; sh = 0 must give the word itself: alignbit by 32 -> use the funnel on (x, y, s & 31)
    const uint32_t l0 = swap ? a.hi : a.lo, h0 = swap ? a.lo : a.hi;           // rotate by 32 first
    const uint32_t s = (32u - sh) & 31u;
    W2 r;
    r.lo = sh ? __builtin_amdgcn_alignbit(l0, h0, s) : l0;
    r.hi = sh ? __builtin_amdgcn_alignbit(h0, l0, s) : h0;
    return r;
}
__device__ __forceinline__ uint64_t keccak_wave_bperm(uint64_t a64, const KL2& K, int l) {
    W2 a{(uint32_t)a64, (uint32_t)(a64 >> 32)};
#pragma unroll
    for (int r = 0; r < 24; r++) {
        W2 as = bperm(K.pi_src, a);
        W2 c = x2(x2(a, bperm(K.th1, a)), x2(x2(bperm(K.th2, a), bperm(K.th3, a)), bperm(K.th4, a)));
        W2 cm = bperm(K.cm_src, c), cp = bperm(K.cp_src, c);
        W2 cr{__builtin_amdgcn_alignbit(cp.lo, cp.hi, 31), __builtin_amdgcn_alignbit(cp.hi, cp.lo, 31)};      // rotl64(cp, 1)
        as = x2(as, x2(cm, cr));
        W2 b = rotl_var(as, K.sh, K.swap);
        W2 b1 = bperm(K.xp1, b), b2 = bperm(K.xp2, b);
        a.lo = b.lo ^ (~b1.lo & b2.lo);
        a.hi = b.hi ^ (~b1.hi & b2.hi);
        if (l == 0) { a.lo ^= (uint32_t)KRC[r]; a.hi ^= (uint32_t)(KRC[r] >> 32); }
    }
    return ((uint64_t)a.hi << 32) | a.lo;
}

// Variant 3: as 1, but the round loop stays a loop (UNROLL rounds per trip): the round constant is loaded at the TOP of the round
// (a scalar load whose latency the round hides) and applied without a branch.
template <int UNROLL>
__device__ __forceinline__ uint64_t keccak_wave_bperm_loop(uint64_t a64, const KL2& K, int l) {
    W2 a{(uint32_t)a64, (uint32_t)(a64 >> 32)};
    const uint32_t m0 = l == 0 ? 0xffffffffu : 0u;
#pragma unroll UNROLL
    for (int r = 0; r < 24; r++) {
        const uint64_t rc = KRC[r];
        W2 as = bperm(K.pi_src, a);
        W2 c = x2(x2(a, bperm(K.th1, a)), x2(x2(bperm(K.th2, a), bperm(K.th3, a)), bperm(K.th4, a)));
        W2 cm = bperm(K.cm_src, c), cp = bperm(K.cp_src, c);
        W2 cr{__builtin_amdgcn_alignbit(cp.lo, cp.hi, 31), __builtin_amdgcn_alignbit(cp.hi, cp.lo, 31)};
        as = x2(as, x2(cm, cr));
        W2 b = rotl_var(as, K.sh, K.swap);
        W2 b1 = bperm(K.xp1, b), b2 = bperm(K.xp2, b);
        a.lo = b.lo ^ (~b1.lo & b2.lo) ^ ((uint32_t)rc & m0);
        a.hi = b.hi ^ (~b1.hi & b2.hi) ^ ((uint32_t)(rc >> 32) & m0);
    }
    return ((uint64_t)a.hi << 32) | a.lo;
}

// LDS variant.  Regions (64-bit words): A2 [0, 128): the state twice (l and l + 25) so that (l + 5 k) needs no modulus; C2 [128, 144):
// the five column parities twice; B2 [144, 144 + 136): rows doubled (10 y + x and 10 y + x + 5) so that x + 1, x + 2 need none.
struct KL3 { int pi_src, cbase, brow; uint32_t sh; bool swap; };
__device__ __forceinline__ void kl3_init(KL3& K, int l) {
    OldLanes k;
    old_lanes_init(k, l);
    const int sx = l < 25 ? k.pi_src % 5 : 0;
    K.pi_src = l < 25 ? k.pi_src : l;
    K.cbase = 128 + sx;              // cm = C2[sx + 4], cp = C2[sx + 1]
    K.brow = 144 + 10 * (l / 5) + l % 5;
    K.sh = (uint32_t)k.rot_src & 31u; K.swap = k.rot_src >= 32;
}
__device__ __forceinline__ uint64_t keccak_wave_lds(uint64_t a, const KL3& K, int l, uint64_t* sh) {
#pragma unroll
    for (int r = 0; r < 24; r++) {
        if (l < 25) { sh[l] = a; sh[l + 25] = a; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint64_t t1 = sh[l + 5], t2 = sh[l + 10], t3 = sh[l + 15], t4 = sh[l + 20];
        uint64_t as = sh[K.pi_src];
        const uint64_t c = a ^ t1 ^ t2 ^ t3 ^ t4;
        if (l < 5) { sh[128 + l] = c; sh[128 + l + 5] = c; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint64_t cm = sh[K.cbase + 4], cp = sh[K.cbase + 1];
        as ^= cm ^ ((cp << 1) | (cp >> 63));
        const W2 bw = rotl_var(W2{(uint32_t)as, (uint32_t)(as >> 32)}, K.sh, K.swap);
        const uint64_t b = ((uint64_t)bw.hi << 32) | bw.lo;
        sh[K.brow] = b; sh[K.brow + 5] = b;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint64_t b1 = sh[K.brow + 1], b2 = sh[K.brow + 2];
        a = b ^ (~b1 & b2);
        if (l == 0) a ^= KRC[r];
    }
    return a;
}

template <int V>
__global__ __launch_bounds__(64) void k(uint64_t* out, const uint64_t* in, int iters, unsigned long long* clk) {
    __shared__ uint64_t sh[144 + 136];
    const int l = threadIdx.x;
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    uint64_t a = l < 25 ? in[l] ^ (uint64_t)blockIdx.x : 0;
    if constexpr (V == 0) {
        OldLanes K;
        old_lanes_init(K, l);
        for (int i = 0; i < iters; i++) a = old_keccak_wave(a, K, l);
    } else if constexpr (V == 6) {
        KeccakLanes K;
        keccak_lanes_init(K, l);
#pragma nounroll
        for (int i = 0; i < iters; i++) a = keccak_f1600_wave(a, K, l);
    } else if constexpr (V == 1) {
        KL2 K;
        kl2_init(K, l);
#pragma nounroll
        for (int i = 0; i < iters; i++) a = keccak_wave_bperm(a, K, l);
    } else if constexpr (V == 3 || V == 4 || V == 5) {
        KL2 K;
        kl2_init(K, l);
#pragma nounroll
        for (int i = 0; i < iters; i++) a = keccak_wave_bperm_loop<V == 3 ? 1 : V == 4 ? 2 : 4>(a, K, l);
    } else {
        KL3 K;
        kl3_init(K, l);
#pragma nounroll
        for (int i = 0; i < iters; i++) a = keccak_wave_lds(a, K, l, sh);
    }
    if (blockIdx.x == 100 && l == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
    if (l < 25) out[(size_t)blockIdx.x * 25 + l] = a;
}

template <int V> int run(const char* name, uint64_t* d_out, const uint64_t* d_in, int blocks, int iters, const uint64_t* want) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    static unsigned long long* d_clk = nullptr;
    if (!d_clk) CHECK(hipMalloc(&d_clk, 16));
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, d_out, d_in, iters, d_clk);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, d_out, d_in, iters, d_clk);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    unsigned long long h_clk[2];
    CHECK(hipMemcpy(h_clk, d_clk, 16, hipMemcpyDeviceToHost));
    double ghz = (double)h_clk[0] / ((double)h_clk[1] * 10.0);
    uint64_t got[25];
    CHECK(hipMemcpy(got, d_out + 100 * 25, sizeof got, hipMemcpyDeviceToHost));
    bool same = memcmp(got, want, sizeof got) == 0;
    double us = ms * 1e3 / iters * (blocks <= 1024 ? 1.0 : 1024.0 / blocks);
    printf("%-44s %8.3f ms  %7.3f us per permutation = %7.0f cycles (%5.0f per round) at %.3f GHz, %d wavefronts, state %s\n", name, ms, us, us * 1e3 * ghz,
           us * 1e3 * ghz / 24, ghz, blocks, same ? "== one-lane reference" : "DIFFERS");
    return same ? 0 : 1;
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 1024, iters = 200;     // 1,024 proofs: one wavefront per SIMD
    uint64_t h_in[25], want[25];
    for (int i = 0; i < 25; i++) h_in[i] = 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1);
    for (int i = 0; i < 25; i++) want[i] = h_in[i] ^ 100ull;
    for (int i = 0; i < iters; i++) keccak_f1600(want);
    uint64_t *d_in, *d_out;
    CHECK(hipMalloc(&d_in, sizeof h_in)); CHECK(hipMalloc(&d_out, (size_t)blocks * 25 * 8));
    CHECK(hipMemcpy(d_in, h_in, sizeof h_in, hipMemcpyHostToDevice));
    int bad = 0;
    bad |= run<0>("0 until round 5 (__shfl, constant at the end)", d_out, d_in, blocks, iters, want);
    bad |= run<1>("1 ds_bpermute, addresses once, unrolled", d_out, d_in, blocks, iters, want);
    bad |= run<2>("2 LDS write2 / read2 on doubled rows", d_out, d_in, blocks, iters, want);
    bad |= run<3>("3 as 1, a loop: constant loaded at the top", d_out, d_in, blocks, iters, want);
    bad |= run<4>("4 as 3, two rounds per trip", d_out, d_in, blocks, iters, want);
    bad |= run<5>("5 as 3, four rounds per trip", d_out, d_in, blocks, iters, want);
    bad |= run<6>("6 hash.h keccak_f1600_wave (the library's)", d_out, d_in, blocks, iters, want);
    return bad;
}
