// Micro-benchmark: k_rv_absorb_V (kernels_verify.h) alone on an idle chip -- the batched verifier's serial floor for many-party
// proofs: 253 STROBE blocks per 1,024-party proof, one wavefront per proof.  Reports the time per block for the whole replay in
// one launch and in the four phases dapol_range_verify_batch uses; tools/ubench_keccak.hip has the permutation alone.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I dapol_amd/csrc -I include tools/ubench_absorb.hip -o build/ubench_absorb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "kernels_verify.h"

using namespace dapol;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 1024, m = argc > 2 ? atoi(argv[2]) : 1024, n = 64;
    VerifyArgs V{};
    V.R.n = n; V.R.m = m; V.R.B = (size_t)B;
    uint32_t* dV;
    VerifyState* dvs;
    CHECK(hipMalloc(&dV, (size_t)B * m * 32));
    CHECK(hipMalloc(&dvs, (size_t)B * sizeof(VerifyState)));
    std::vector<uint32_t> hv((size_t)B * m * 8);
    uint32_t x = 12345;
    for (auto& w : hv) { x = x * 1664525u + 1013904223u; w = x; }
    CHECK(hipMemcpy(dV, hv.data(), hv.size() * 4, hipMemcpyHostToDevice));
    V.R.Vc = dV;
    V.vs = dvs;
    const VHead H = rv_transcript_head(n, m);
    const uint32_t blocks = (H.pos + 41u * (uint32_t)m) / 166u;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int phases : {1, 4}) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            CHECK(hipEventRecord(e0));
            for (int k = 0; k < phases; k++) hipLaunchKernelGGL(k_rv_absorb_V, dim3(B), dim3(64), 0, 0, V, H, k * (m / phases), (k + 1) * (m / phases));
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("k_rv_absorb_V  %d proofs x %d parties, %u blocks, %d phase(s): %.3f ms = %.2f us per block\n", B, m, blocks, phases, best, best * 1e3 / blocks);
    }
    std::vector<uint64_t> st(25);
    std::vector<uint8_t> raw(sizeof(VerifyState));
    CHECK(hipMemcpy(raw.data(), dvs + (B - 1), sizeof(VerifyState), hipMemcpyDeviceToHost));
    const VerifyState* vs = reinterpret_cast<const VerifyState*>(raw.data());
    // the one-lane reference of the same stream
    Strobe s;
    merlin_init(s, LBL_APP_TRANSCRIPT);
    merlin_append_bytes(s, LBL_DOM_SEP, LBL_RANGEPROOF_DOMAIN);
    merlin_append_u64(s, LBL_N, (uint64_t)n);
    merlin_append_u64(s, LBL_M, (uint64_t)m);
    for (int j = 0; j < m; j++) merlin_append_words(s, LBL_V, hv.data() + ((size_t)(B - 1) * m + j) * 8, 8);
    bool same = s.pos == vs->st_pos && s.pos_begin == vs->st_pos_begin;
    for (int i = 0; i < 25; i++) same &= s.s[i] == vs->st[i];
    printf("state of the last proof %s the one-lane replay of the same commitments\n", same ? "==" : "DIFFERS FROM");
    return same ? 0 : 1;
}
