// Micro-benchmark: does the ORDER in which a fixed-base MSM walks the window tables matter on MI355X?
// The same table additions (ge_madd over 128-byte affine-niels entries of a W-bit window table the size of the product's,
// 4,096 rows), in three schedules:
//   ps    proof-stationary (what k_rp_msm<0, 2> does): a lane owns 1,024 terms of one proof's list and walks
//         window-outer / term-inner with W shared doublings per window; every lookup is a random line of a random row.
//   gs    generator-stationary, accumulators in registers: a lane owns ONE (proof, window) accumulator, every wavefront of
//         the chip sweeps the rows in the same order, so the row being read (2^(W-1)+1 lines) is shared by all resident
//         lanes and sits in the Infinity Cache; no doublings in the loop (window sums are combined afterwards).
//   gst   generator-stationary in tiles of R rows with the accumulators carried in HBM between tiles (more lanes than the
//         chip holds, so a row is read from HBM once per MSM instead of once per resident set).
// Prints the SIMD time per wave-addition (doublings charged to the additions for ps) and the shader clock.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I dapol_amd/csrc tools/ubench_msm_order.hip -o build/ubench_msm_order
// Run:   build/ubench_msm_order [W=17] [rows=4096] [which=ps,gs,gs2,gst] [occ-limited lanes...]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include "ge.h"

using namespace dapol;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return (uint32_t)x;
}
// entries: 27 reduced limbs (29 bits; limbs 8, 17, 26 hold 23 bits) + 5 words of padding
__global__ void k_fill_table(int32_t* t, size_t words) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    for (; i < words; i += step) {
        int k = (int)(i & 31);
        uint32_t r = mix(i);
        t[i] = k >= 27 ? 0 : (int32_t)(r & ((k % 9) == 8 ? 0x7fffffu : 0x1fffffffu));
    }
}
__global__ void k_fill_digits(int32_t* d, size_t n, int wbits, uint64_t salt) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    const int half = 1 << (wbits - 1);
    for (; i < n; i += step) d[i] = (int32_t)(mix(i ^ salt) % (uint32_t)(2 * half + 1)) - half;
}
__global__ void k_fill_acc(int32_t* a, size_t lanes) {       // SoA: word k of lane l at a[k * lanes + l]
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lanes) return;
    for (int k = 0; k < 36; k++) a[(size_t)k * lanes + i] = (int32_t)(mix(i * 36 + k + 99) & ((k % 9) == 8 ? 0x7fffffu : 0x1fffffffu));
}

__device__ __forceinline__ void load_entry(ge_niels& q, const int32_t* e) {
    const v4i* p = reinterpret_cast<const v4i*>(e);
    v4i a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3], a4 = p[4], a5 = p[5], a6 = p[6];
    q.ypx.v[0] = a0.x; q.ypx.v[1] = a0.y; q.ypx.v[2] = a0.z; q.ypx.v[3] = a0.w;
    q.ypx.v[4] = a1.x; q.ypx.v[5] = a1.y; q.ypx.v[6] = a1.z; q.ypx.v[7] = a1.w;
    q.ypx.v[8] = a2.x;
    q.ymx.v[0] = a2.y; q.ymx.v[1] = a2.z; q.ymx.v[2] = a2.w;
    q.ymx.v[3] = a3.x; q.ymx.v[4] = a3.y; q.ymx.v[5] = a3.z; q.ymx.v[6] = a3.w;
    q.ymx.v[7] = a4.x; q.ymx.v[8] = a4.y;
    q.xy2d.v[0] = a4.z; q.xy2d.v[1] = a4.w;
    q.xy2d.v[2] = a5.x; q.xy2d.v[3] = a5.y; q.xy2d.v[4] = a5.z; q.xy2d.v[5] = a5.w;
    q.xy2d.v[6] = a6.x; q.xy2d.v[7] = a6.y; q.xy2d.v[8] = a6.z;
}
// 96-byte packed entries (round 5, lever b): 3 x 255 bits in 24 words, six 16-byte loads; unpacking = per field element eight
// funnel shifts + nine masks (one v_alignbit + one v_and per limb).  The stride is 96 B, so two entries of four straddle a 128-byte line.
__device__ __forceinline__ void unpack255(fe& r, const uint32_t* w) {     // w[0..7]: 255 bits little-endian -> nine 29-bit limbs
#pragma unroll
    for (int i = 0; i < FE_NL; i++) {
        const int bit = 29 * i, k = bit >> 5, sh = bit & 31;
        uint32_t lo = w[k] >> sh;
        if (sh > 3 && k + 1 < 8) lo |= w[k + 1] << (32 - sh);
        r.v[i] = (int32_t)(lo & (i == 8 ? 0x7fffffu : 0x1fffffffu));
    }
}
__device__ __forceinline__ void load_entry_packed(ge_niels& q, const int32_t* e) {
    const v4i* p = reinterpret_cast<const v4i*>(e);
    v4i a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3], a4 = p[4], a5 = p[5];
    uint32_t w[24] = {(uint32_t)a0.x, (uint32_t)a0.y, (uint32_t)a0.z, (uint32_t)a0.w, (uint32_t)a1.x, (uint32_t)a1.y, (uint32_t)a1.z, (uint32_t)a1.w,
                      (uint32_t)a2.x, (uint32_t)a2.y, (uint32_t)a2.z, (uint32_t)a2.w, (uint32_t)a3.x, (uint32_t)a3.y, (uint32_t)a3.z, (uint32_t)a3.w,
                      (uint32_t)a4.x, (uint32_t)a4.y, (uint32_t)a4.z, (uint32_t)a4.w, (uint32_t)a5.x, (uint32_t)a5.y, (uint32_t)a5.z, (uint32_t)a5.w};
    unpack255(q.ypx, w); unpack255(q.ymx, w + 8); unpack255(q.xy2d, w + 16);
}
__device__ __forceinline__ void acc_init(ge_p3& acc, int seed) {
    for (int i = 0; i < FE_NL; i++) {
        acc.X.v[i] = (seed * 7 + i) & 0xffff; acc.Y.v[i] = (seed * 11 + i) & 0xffff; acc.Z.v[i] = (seed * 13 + i) & 0xffff; acc.T.v[i] = (seed * 17 + i) & 0xffff;
    }
}
__device__ __forceinline__ void acc_store(int32_t* o, const ge_p3& acc) {
    int x = 0;
    for (int i = 0; i < FE_NL; i++) x ^= acc.X.v[i] ^ acc.Y.v[i] ^ acc.Z.v[i] ^ acc.T.v[i];
    *o = x;
}

// ---------------------------------------------------------------------------------------------- proof-stationary
// 16 proofs per wavefront, 2 lanes per list (k_rp_msm<0, 2>): lane = (proof sub, side, ql); 1,024 terms per lane per window in
// runs of four; digits dig[proof][nwin][4096] (positions as in the product: 64 * (q / 32) + q % 32 + 32 * side).
__global__ __launch_bounds__(64, 4) void k_ps(int32_t* out, const int32_t* tbl, const int32_t* dig, int wbits, int nwin, int n_rows, size_t row_words,
                                              unsigned long long* clk) {
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    const int l = threadIdx.x, sub = l >> 2, side = (l >> 1) & 1, ql = l & 1;
    const size_t b = (size_t)blockIdx.x * 16 + sub;
    const int N = n_rows / 2;                    // terms per list
    const int32_t* dg = dig + b * (size_t)nwin * 4096 + 32 * side;
    ge_p3 acc;
    acc_init(acc, l);
    for (int w = nwin - 1; w >= 0; w--) {
        if (w != nwin - 1)
            for (int d = 0; d < wbits; d++) ge_dbl(acc, acc, d == wbits - 1);
        const int32_t* dw = dg + (size_t)w * 4096;
        v4i d4 = {0, 0, 0, 0};
#pragma nounroll
        for (int it = 0; it < N / 2; it++) {
            int q = 4 * (2 * (it >> 2) + ql) + (it & 3);
            if ((it & 3) == 0) d4 = *reinterpret_cast<const v4i*>(dw + 64 * (q >> 5) + (q & 31));
            const int d = d4.x;
            d4.x = d4.y; d4.y = d4.z; d4.z = d4.w;
            const int row = side * N + q;
            const int ad = d < 0 ? -d : d;
            ge_niels e;
            load_entry(e, tbl + (uint64_t)(uint32_t)row * (uint32_t)row_words + (uint32_t)(ad * 32));
            ge_madd(acc, acc, e, d < 0);
        }
    }
    if (blockIdx.x == 100 && l == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
    acc_store(out + (size_t)blockIdx.x * 64 + l, acc);
}

// ------------------------------------------------------------------------------------------ generator-stationary
// NACC accumulators per lane (1: occupancy 4; 2: occupancy 3).  Lane id -> accumulator(s) a * lanes + lane; digits dig[row][NACC * lanes].
// carry != 0: accumulators are loaded from / stored to `accs` (SoA) -- the tile form.
template <int NACC, int OCC, int DV = 0, int PK = 0>
__global__ __launch_bounds__(64, OCC) void k_gs(int32_t* out, int32_t* accs, const int32_t* tbl, const int32_t* dig, size_t lanes, int row0, int rows, size_t row_words,
                                                int carry, unsigned long long* clk) {
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    const size_t lane = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (lane >= lanes) return;
    ge_p3 acc[NACC];
    for (int a = 0; a < NACC; a++) {
        if (carry) {
            const int32_t* p = accs + (size_t)a * 36 * lanes + lane;
            for (int i = 0; i < FE_NL; i++) {
                acc[a].X.v[i] = p[(size_t)i * lanes]; acc[a].Y.v[i] = p[(size_t)(9 + i) * lanes];
                acc[a].Z.v[i] = p[(size_t)(18 + i) * lanes]; acc[a].T.v[i] = p[(size_t)(27 + i) * lanes];
            }
        } else acc_init(acc[a], threadIdx.x + a);
    }
    const int32_t* dg = dig + lane;
    v4i d4 = {0, 0, 0, 0};
#pragma nounroll
    for (int g = 0; g < rows; g++) {
        const int32_t* trow = tbl + (uint64_t)(uint32_t)(row0 + g) * (uint32_t)row_words;
        if constexpr (DV) {
            if ((g & 3) == 0) d4 = reinterpret_cast<const v4i*>(dig)[(size_t)(g >> 2) * lanes + lane];
        }
#pragma unroll
        for (int a = 0; a < NACC; a++) {
            int d;
            if constexpr (DV) { d = d4.x; d4.x = d4.y; d4.y = d4.z; d4.z = d4.w; }
            else d = dg[((size_t)g * NACC + a) * lanes];
            const int ad = d < 0 ? -d : d;
            ge_niels e;
            if constexpr (PK) load_entry_packed(e, trow + (uint32_t)(ad * 24));
            else load_entry(e, trow + (uint32_t)(ad * 32));
            ge_madd(acc[a], acc[a], e, d < 0);
        }
    }
    if (blockIdx.x == 100 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
    if (carry) {
        for (int a = 0; a < NACC; a++) {
            int32_t* p = accs + (size_t)a * 36 * lanes + lane;
            for (int i = 0; i < FE_NL; i++) {
                p[(size_t)i * lanes] = acc[a].X.v[i]; p[(size_t)(9 + i) * lanes] = acc[a].Y.v[i];
                p[(size_t)(18 + i) * lanes] = acc[a].Z.v[i]; p[(size_t)(27 + i) * lanes] = acc[a].T.v[i];
            }
        }
    } else {
        for (int a = 0; a < NACC; a++) acc_store(out + (size_t)a * lanes + lane, acc[a]);
    }
}

// Tile form with the NEXT row's entry requested in the MIDDLE of the addition: the three products that consume the entry come
// first, then the loads of the next entry go out into the same registers, then the four products that finish the addition.
__device__ __forceinline__ void madd_first(fe& E, fe& F, fe& G, fe& H, const ge_p3& p, const ge_niels& q, bool neg) {
    fe ypx, ymx, A, B, C, D, qa = q.ymx, qb = q.ypx;
    fe_cswap(qa, qb, neg);
    fe_add(ypx, p.Y, p.X);
    fe_sub(ymx, p.Y, p.X);
    fe_mul(A, ymx, qa);
    fe_mul(B, ypx, qb);
    fe_mul(C, p.T, q.xy2d);
    fe nC;
    fe_neg(nC, C);
    fe_cmov(C, nC, neg);
    fe_add(D, p.Z, p.Z);
    fe_sub(E, B, A);
    fe_add(H, B, A);
    fe_sub(F, D, C);
    fe_add(G, D, C);
    fe_carry(G, G);
}
__device__ __forceinline__ void madd_second(ge_p3& r, const fe& E, const fe& F, const fe& G, const fe& H) {
    fe_mul(r.X, F, E);
    fe_mul(r.Y, H, G);
    fe_mul(r.Z, F, G);
    fe_mul(r.T, H, E);
}
__global__ __launch_bounds__(64, 3) void k_gs_mid(int32_t* accs, const int32_t* tbl, const int32_t* dig, size_t lanes, int row0, int rows, size_t row_words, unsigned long long* clk) {
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    const size_t lane = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (lane >= lanes) return;
    ge_p3 acc;
    int32_t* p = accs + lane;
    for (int i = 0; i < FE_NL; i++) {
        acc.X.v[i] = p[(size_t)i * lanes]; acc.Y.v[i] = p[(size_t)(9 + i) * lanes];
        acc.Z.v[i] = p[(size_t)(18 + i) * lanes]; acc.T.v[i] = p[(size_t)(27 + i) * lanes];
    }
    const v4i* dg = reinterpret_cast<const v4i*>(dig) + lane;
    v4i d4 = dg[0];
    ge_niels e;
    int d = d4.x;
    load_entry(e, tbl + (uint64_t)(uint32_t)row0 * (uint32_t)row_words + (uint32_t)((d < 0 ? -d : d) * 32));
#pragma nounroll
    for (int g = 0; g < rows; g++) {
        fe E, F, G, H;
        madd_first(E, F, G, H, acc, e, d < 0);
        d4.x = d4.y; d4.y = d4.z; d4.z = d4.w;
        if (((g + 1) & 3) == 0 && g + 1 < rows) d4 = dg[(size_t)((g + 1) >> 2) * lanes];
        d = d4.x;
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < rows) load_entry(e, tbl + (uint64_t)(uint32_t)(row0 + g + 1) * (uint32_t)row_words + (uint32_t)((d < 0 ? -d : d) * 32));
        __builtin_amdgcn_sched_barrier(0);
        madd_second(acc, E, F, G, H);
    }
    if (blockIdx.x == 100 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
    for (int i = 0; i < FE_NL; i++) {
        p[(size_t)i * lanes] = acc.X.v[i]; p[(size_t)(9 + i) * lanes] = acc.Y.v[i];
        p[(size_t)(18 + i) * lanes] = acc.Z.v[i]; p[(size_t)(27 + i) * lanes] = acc.T.v[i];
    }
}

static double clock_ghz(unsigned long long* d_clk) {
    unsigned long long h[2];
    CHECK(hipMemcpy(h, d_clk, 16, hipMemcpyDeviceToHost));
    return h[1] ? (double)h[0] / ((double)h[1] * 10.0) : 0.0;
}

int main(int argc, char** argv) {
    int W = argc > 1 ? atoi(argv[1]) : 17;
    int n_rows = argc > 2 ? atoi(argv[2]) : 4096;
    std::string which = argc > 3 ? argv[3] : "ps,gs,gs2,gst";
    const double secs = argc > 4 ? atof(argv[4]) : 2.0;       // run each variant about this long (the clock settles under the power cap)
    const int nwin = 253 / W + 1;
    const size_t entries = ((size_t)1 << (W - 1)) + 1, row_words = entries * 32;
    const size_t tbl_words = row_words * n_rows;
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int simds = prop.multiProcessorCount * 4;
    printf("# W=%d nwin=%d rows=%d table=%.1f GB  CUs=%d\n", W, nwin, n_rows, tbl_words * 4 / 1e9, prop.multiProcessorCount);
    int32_t* tbl; CHECK(hipMalloc(&tbl, tbl_words * 4));
    hipLaunchKernelGGL(k_fill_table, dim3(65536), dim3(256), 0, 0, tbl, tbl_words);
    CHECK(hipDeviceSynchronize());
    unsigned long long* d_clk; CHECK(hipMalloc(&d_clk, 16)); CHECK(hipMemset(d_clk, 0, 16));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto has = [&](const char* s) { return (("," + which + ",").find(std::string(",") + s + ",") != std::string::npos); };

    if (has("ps")) {
        const size_t B = 65536;                       // one chunk of the product: 4,096 wavefronts of 16 proofs
        const size_t ndig = B * nwin * 4096;
        int32_t* dig; CHECK(hipMalloc(&dig, ndig * 4));
        int32_t* out; CHECK(hipMalloc(&out, B * 4 * 4));
        hipLaunchKernelGGL(k_fill_digits, dim3(65536), dim3(256), 0, 0, dig, ndig, W, 1ull);
        CHECK(hipDeviceSynchronize());
        const int terms = n_rows / 4;                  // per lane per window
        const double adds_per_launch = (double)B * 4 * terms * nwin;
        int reps = 1; float ms = 0;
        for (int pass = 0; pass < 2; pass++) {
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < reps; r++)
                hipLaunchKernelGGL(k_ps, dim3(B / 16), dim3(64), 0, 0, out, tbl, dig, W, nwin, n_rows, row_words, d_clk);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (pass == 0) { reps = (int)(secs * 1000.0 / ms) + 1; }
        }
        double per_launch = ms / reps;
        double ns_wave_add = per_launch * 1e6 * simds / (adds_per_launch / 64);
        printf("ps   B=%zu  %8.2f ms per launch (%d reps)  %7.1f ns SIMD time per wave-addition (doublings included)  clock %.3f GHz\n", B, per_launch, reps, ns_wave_add, clock_ghz(d_clk));
        CHECK(hipFree(dig)); CHECK(hipFree(out));
    }
    auto run_gs = [&](const char* name, int nacc, size_t lanes, int tile_rows, int dv = 0) {
        // one list = n_rows / 2 rows; the two lists of an MSM are two sweeps (rows [0, n/2) and [n/2, n))
        const int list_rows = n_rows / 2;
        const size_t ndig = (size_t)list_rows * nacc * lanes;
        int32_t* dig; CHECK(hipMalloc(&dig, ndig * 4));
        int32_t* out; CHECK(hipMalloc(&out, (size_t)nacc * lanes * 4));
        int32_t* accs = nullptr;
        if (tile_rows) { CHECK(hipMalloc(&accs, (size_t)nacc * lanes * 36 * 4)); hipLaunchKernelGGL(k_fill_acc, dim3((nacc * lanes + 255) / 256), dim3(256), 0, 0, accs, nacc * lanes); }
        hipLaunchKernelGGL(k_fill_digits, dim3(65536), dim3(256), 0, 0, dig, ndig, W, 2ull);
        CHECK(hipDeviceSynchronize());
        const double adds_per_sweep = (double)list_rows * nacc * lanes;
        const int blocks = (int)((lanes + 63) / 64);
        int reps = 1; float ms = 0;
        for (int pass = 0; pass < 2; pass++) {
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < reps; r++) {
                const int row_base = (r & 1) * list_rows;
                if (!tile_rows) {
                    if (nacc == 1) hipLaunchKernelGGL((k_gs<1, 4>), dim3(blocks), dim3(64), 0, 0, out, accs, tbl, dig, lanes, row_base, list_rows, row_words, 0, d_clk);
                    else hipLaunchKernelGGL((k_gs<2, 3>), dim3(blocks), dim3(64), 0, 0, out, accs, tbl, dig, lanes, row_base, list_rows, row_words, 0, d_clk);
                } else {
                    for (int t = 0; t < list_rows; t += tile_rows) {
                        const size_t rw = dv == 4 ? row_words / 32 * 24 : row_words;       // packed: rows of 96-byte entries
                        if (dv == 3) hipLaunchKernelGGL((k_gs<1, 5, 1>), dim3(blocks), dim3(64), 0, 0, out, accs, tbl, dig + (size_t)t * lanes, lanes, row_base + t, tile_rows, row_words, 1, d_clk);
                        else if (dv == 4) hipLaunchKernelGGL((k_gs<1, 4, 1, 1>), dim3(blocks), dim3(64), 0, 0, out, accs, tbl, dig + (size_t)t * lanes, lanes, row_base + t, tile_rows, rw, 1, d_clk);
                        else if (dv == 2) hipLaunchKernelGGL(k_gs_mid, dim3(blocks), dim3(64), 0, 0, accs, tbl, dig + (size_t)t * lanes, lanes, row_base + t, tile_rows, row_words, d_clk);
                        else if (dv) hipLaunchKernelGGL((k_gs<1, 4, 1>), dim3(blocks), dim3(64), 0, 0, out, accs, tbl, dig + (size_t)t * lanes, lanes, row_base + t, tile_rows, row_words, 1, d_clk);
                        else hipLaunchKernelGGL((k_gs<1, 4>), dim3(blocks), dim3(64), 0, 0, out, accs, tbl, dig + (size_t)t * lanes, lanes, row_base + t, tile_rows, row_words, 1, d_clk);
                    }
                }
            }
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (pass == 0) reps = (int)(secs * 1000.0 / ms) + 1;
        }
        double per_sweep = ms / reps;
        double ns_wave_add = per_sweep * 1e6 * simds / (adds_per_sweep / 64);
        printf("%-4s lanes=%zu x %d acc, tile=%d  %8.2f ms per list sweep (%d reps)  %7.1f ns SIMD time per wave-addition  clock %.3f GHz\n", name, lanes, nacc, tile_rows,
               per_sweep, reps, ns_wave_add, clock_ghz(d_clk));
        CHECK(hipFree(dig)); CHECK(hipFree(out)); if (accs) CHECK(hipFree(accs));
    };
    const size_t resident4 = (size_t)simds * 4 * 64, resident3 = (size_t)simds * 3 * 64;
    if (has("gs")) run_gs("gs", 1, resident4, 0);
    if (has("gsh")) run_gs("gsh", 1, resident4 / 2, 0);          // half the chip's residency: does the sweep still hit?
    if (has("gs2")) run_gs("gs2", 2, resident3, 0);
    if (has("gsx")) run_gs("gsx", 1, resident4 * 4, 0);           // four rounds of resident wavefronts in one launch (no alignment between rounds)
    if (has("gst")) { run_gs("gst", 1, (size_t)65536 * nwin, 16); run_gs("gst", 1, (size_t)65536 * nwin, 8); }
    if (has("gst32")) run_gs("gst", 1, (size_t)65536 * nwin, 32);
    // explicit specs: gst:<lanes>:<tile rows>[:v]   (v = digits of four rows per 16-byte load)
    {
        std::string w = which + ",";
        size_t pos = 0;
        while (true) {
            size_t c = w.find(',', pos);
            if (c == std::string::npos) break;
            std::string tok = w.substr(pos, c - pos);
            pos = c + 1;
            if (tok.rfind("gst:", 0) == 0) {
                size_t lanes = 0; int tile = 16; char v = 0;
                sscanf(tok.c_str(), "gst:%zu:%d:%c", &lanes, &tile, &v);
                // w = as v with five wavefronts per SIMD (96 VGPRs); p = as v over 96-byte packed entries
                run_gs(v == 'v' ? "gstv" : v == 'm' ? "gstm" : v == 'w' ? "gstw" : v == 'p' ? "gstp" : "gst", 1, lanes, tile,
                       v == 'v' ? 1 : v == 'm' ? 2 : v == 'w' ? 3 : v == 'p' ? 4 : 0);
            }
        }
    }
    return 0;
}
