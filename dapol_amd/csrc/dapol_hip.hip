// libdapol_hip.so -- C ABI (include/dapol_hip.h) over the gfx950 kernels.  Host side only orchestrates: every
// byte of arithmetic (generator derivation included) runs on the GPU; there is no CPU fallback.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>          // radix sort only (plain library plumbing for the leaf-derivation sorts)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/dapol_hip.h"
#include "kernels_ctx_tree.h"
#include "kernels_range.h"
#include "kernels_range_gs.h"
#include "kernels_verify.h"
#include "kernels_leaf.h"

using namespace dapol;

static thread_local std::string g_last_error;
static int32_t fail_hip(hipError_t e, const char* what, int line) {
    char buf[256];
    snprintf(buf, sizeof buf, "%s failed at dapol_hip.hip:%d: %s", what, line, hipGetErrorString(e));
    g_last_error = buf;
    // Several entry points fork work onto the context's side streams (chunks in flight, the verifier's own-point ladders, the
    // leaves' commitments of a small tree) and join it later; an error return in between must not leave that work running on the
    // context's scratch, which the next call re-uses without any ordering against those streams.  Errors are rare: drain the device.
    (void)hipDeviceSynchronize();
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? DAPOL_ERR_OUT_OF_MEMORY : DAPOL_ERR_HIP;
}
static int32_t fail(int32_t code, const char* msg) {
    g_last_error = msg;
    return code;
}
#define HIPCHK(x)                                                  \
    do {                                                           \
        hipError_t e_ = (x);                                       \
        if (e_ != hipSuccess) return fail_hip(e_, #x, __LINE__);   \
    } while (0)
#define LAUNCH_CHECK() HIPCHK(hipGetLastError())

// The DAPOL_* environment variables are measurement knobs (A/B switches of tools/ and tests/).  They are read only when the
// process has opted in -- DAPOL_ENV_KNOBS set to anything but "0", or dapol_env_knobs(1) -- so that a host embedding the library
// does not inherit behaviour from stray variables; what an embedder may want to set is a field of dapol_options.
static std::atomic<int> g_env_knobs{-1};          // -1: ask the environment
static std::atomic<unsigned long long> g_fork_guard_waits{0};   // side streams a ForkGuard had to wait for (early returns between fork and join)
static const char* knob(const char* name) {
    int on = g_env_knobs.load(std::memory_order_relaxed);
    if (on < 0) {
        const char* e = getenv("DAPOL_ENV_KNOBS");
        on = (e && *e && strcmp(e, "0") != 0) ? 1 : 0;
    }
    return on ? getenv(name) : nullptr;
}
// Fault injection and limit overrides that exist for tests/ only (DAPOL_TEST_FAIL_AFTER_FORK, DAPOL_TEST_FAIL_UPDATE_MIDWAY,
// DAPOL_LEAF_MAX_TRIES: they make healthy calls fail, or change when DapolError::FailedToMapIndex fires) need a SECOND opt-in,
// DAPOL_TEST_HOOKS=1, read once per process: the measurement scripts of tools/ export DAPOL_ENV_KNOBS alone and can never trip them,
// and a call site costs a flag test instead of a getenv + strcmp (round-4 advisor).
static const char* test_knob(const char* name) {
    static const bool hooks = [] { const char* e = getenv("DAPOL_TEST_HOOKS"); return e && !strcmp(e, "1"); }();
    return hooks ? knob(name) : nullptr;
}
int32_t dapol_env_knobs(int32_t enable) {
    int old = g_env_knobs.exchange(enable ? 1 : 0);
    if (old < 0) { const char* e = getenv("DAPOL_ENV_KNOBS"); old = (e && *e && strcmp(e, "0") != 0) ? 1 : 0; }
    return old;
}

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; return *this; }
    ~DevBuf() { release(); }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
    hipError_t alloc(size_t count) {
        release();
        n = count;
        if (count == 0) return hipSuccess;
        return hipMalloc((void**)&p, count * sizeof(T));
    }
};

static inline unsigned nblk(size_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

#include "wire_scope.inc"

// ------------------------------------------------------------------------------------------------ context
struct dapol_ctx {
    // One reference for the caller's handle plus one per tree / workload built on the context: dapol_ctx_destroy only
    // drops the caller's, so handles may be destroyed in any order (garbage-collected language bindings do exactly that).
    std::atomic<int> refs{1};
    dapol_options opt{};                                 // zeros = the library's own choices (dapol_ctx_set_options)
    int device = 0;
    int max_parties = 0;
    int n_cu = 256;                                      // hipDeviceProp_t::multiProcessorCount (MI355X: 256)
    int msm_waves_per_cu = 4 * DAPOL_MSM_OCC;            // resident wavefronts of k_rp_msm per CU (occupancy API, dapol_ctx_create)
    unsigned msm_dyn_lds = 0;                            // dynamic LDS bytes of the MSM launches: caps the residency (DAPOL_MSM_OCC_CAP)
    // wavefronts the dominant kernel keeps resident on the chip: launches are sized in whole rounds of this many
    size_t resident_waves() const { return (size_t)n_cu * (size_t)msm_waves_per_cu; }
    hipStream_t stream = nullptr;
    hipStream_t side[3] = {nullptr, nullptr, nullptr};   // further pipelines of the range prover (several chunks in flight)
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_v[4] = {nullptr, nullptr, nullptr, nullptr};   // the verifier's commitments landing in column blocks (host_verify.inc: VArrival)
    // Opt-in experiment (DAPOL_MSM_SERIAL=1, measured slower): the generator-stationary MSMs of ALL chunks in flight through ONE
    // stream, one sweep at a time, the other chunks' scalar kernels beside it.  ev_msm_pre / _post[chunk lane] order a chunk's own
    // stream around its MSMs.
    hipStream_t msm_stream = nullptr;
    hipEvent_t ev_msm_pre[4] = {nullptr, nullptr, nullptr, nullptr}, ev_msm_post[4] = {nullptr, nullptr, nullptr, nullptr};
    // Measurement knob DAPOL_STREAM_LAYOUT (round 4, profiles/r07*_stream_layout*.txt): streams with CU masks / priorities, made on
    // first use.  layout_lane[] carry the chunks, layout_msm the VALU-bound launches of all of them when the layout separates those.
    std::string layout_name;
    hipStream_t layout_lane[2] = {nullptr, nullptr}, layout_msm = nullptr;
    hipEvent_t layout_ev[2] = {nullptr, nullptr};
    bool layout_split_msm = false;
    DevBuf<int32_t> table;       // window tables
    DevBuf<uint32_t> gens_comp;  // compressed base points of every row (for dapol_ctx_generator)
    TableView tv{};
    RangeScratch scratch;        // grown on demand by the range prover
    RangeScratch vio;            // dapol_range_verify_batch's device copies of the caller's proofs / commitments / verdicts: kept between calls
                                 // (a 34 MB hipMalloc + hipFree per call is a few hundred microseconds of a 7 ms pass; at most 1 GB is kept)
    // LANES for the sub-proofs of ONE small call (round 6, host_range.inc: prove_policy_device): a policy's plan with several groups
    // of sub-proofs -- splitting at height 24 is a 16-party and an 8-party proof, benches/dapol.rs:71-78 -- is latency-bound, and its
    // groups are independent statements; each extra group runs on a lane of its own: a shallow context that shares the tables (tv)
    // and owns its streams, events and scratch.  Made on first use, freed with the context.
    dapol_ctx* aux[3] = {nullptr, nullptr, nullptr};
    uint32_t* h_pinned = nullptr;    // 16 page-locked words: a device-to-host copy into pageable memory blocks the host until the stream has
                                     // drained, which would serialise the lanes; into this it is queued like a kernel
};

// Width of the context's node hash D: 8 words (BLAKE3, Blake2s) or 16 (Blake2b).  Every H buffer of the C ABI holds this many
// bytes per node (dapol_ctx_digest_bytes).
static inline int ctx_hw(const dapol_ctx* c) { return dg_hash_words(c->tv.digest); }
static inline size_t ctx_hash_bytes(const dapol_ctx* c) { return (size_t)ctx_hw(c) * 4; }
static inline bool ctx_wide(const dapol_ctx* c) { return ctx_hw(c) == 16; }
// Paths that exist for the 32-byte digests only (the sharded / workload / record paths and Dapol::new's leaf derivation, which the
// reference itself refuses for other sizes, src/dapol/mod.rs:101-103).
#define NEEDS_32_BYTE_DIGEST(ctx, what)                                                                                                  \
    do {                                                                                                                                 \
        if (ctx_wide(ctx)) return fail(DAPOL_ERR_INVALID_DIGEST_SIZE, what " needs a 32-byte node digest (DapolError::InvalidDigestSize)"); \
    } while (0)
int32_t dapol_ctx_digest_bytes(dapol_ctx* ctx, int32_t* bytes) {
    if (!ctx || !bytes) return fail(DAPOL_ERR_INVALID_ARGUMENT, "null argument");
    *bytes = (int32_t)ctx_hash_bytes(ctx);
    return DAPOL_OK;
}

const char* dapol_strerror(int32_t code) {
    switch (code) {
        case DAPOL_OK: return "ok";
        case DAPOL_ERR_TREE_HEIGHT_TOO_BIG: return "DAPOL tree height must not exceed 64";
        case DAPOL_ERR_SPARSITY_TOO_SMALL: return "tree height too small for the liability set (2^height < 2n)";
        case DAPOL_ERR_INVALID_DIGEST_SIZE: return "digest size must be 32 bytes";
        case DAPOL_ERR_DUPLICATED_INTERNAL_ID: return "liability set contains a duplicated internal ID";
        case DAPOL_ERR_FAILED_TO_MAP_INDEX: return "failed to map audit ID to a tree index within 128 tries";
        case DAPOL_ERR_BYTES_NOT_ENOUGH: return "decoding: bytes not enough";
        case DAPOL_ERR_VALUE_DECODING: return "decoding: value decoding error";
        case DAPOL_ERR_INVALID_ARGUMENT: return "invalid argument";
        case DAPOL_ERR_UNKNOWN_LEAF: return "no liability at the requested leaf";
        case DAPOL_ERR_NO_DEVICE: return "no usable HIP device (the proving path has no CPU fallback)";
        case DAPOL_ERR_HIP: return "HIP runtime error";
        case DAPOL_ERR_OUT_OF_MEMORY: return "out of device memory";
        case DAPOL_ERR_COMM: return "RCCL error";
        default: return "unknown status";
    }
}
const char* dapol_last_error(void) { return g_last_error.c_str(); }

static bool options_ok(const dapol_options* o) {
    if (!o) return true;
    if (o->struct_size != 0 && o->struct_size != (int32_t)sizeof(dapol_options)) return false;
    if (o->window_bits && (o->window_bits < WBITS_MIN || o->window_bits > WBITS_MAX)) return false;
    if (o->gs_tile_rows && (o->gs_tile_rows < 4 || o->gs_tile_rows % 4)) return false;
    if (o->streams < 0 || o->streams > 4 || o->chunk_proofs < 0 || o->table_gb < 0 || o->scratch_gb < 0) return false;
    if (o->tail_length && o->tail_length != -1 && o->tail_length != 32 && o->tail_length != 64 && o->tail_length != 128 && o->tail_length != 256) return false;
    if (o->small_call_max < 0 || o->verify_batch_min < 0 || o->update_incremental_max < -1) return false;
    if (o->gs_slices != 0 && o->gs_slices != 1 && o->gs_slices != 2 && o->gs_slices != 4 && o->gs_slices != 8 && o->gs_slices != 16) return false;
    if (o->profile != DAPOL_PROFILE_BENCH && o->profile != DAPOL_PROFILE_HOST) return false;
    return true;
}
// DAPOL_STREAM_LAYOUT=<name> (measurement knob): where the chunks in flight and their VALU-bound launches run.
//   split_xcd     two chunks in flight, each on its own four XCDs (128 CUs, four L2s)
//   split_cu      two chunks in flight, each on 16 CUs of every XCD
//   msm:<K>       the VALU-bound launches of all chunks (sweeps, materialisation, tail MSM and tables) one after the other on 256 - K
//                 CUs, everything else (streaming scalar kernels, Fiat-Shamir) on the other K CUs (K / 8 per XCD)
//   prio          no masks: the VALU-bound launches on a LOW-priority stream, one after the other, the rest on HIGH-priority streams
//   prio_lanes    no masks, no separation: two chunks in flight on two streams as by default, but created with HIGH priority for
//                 the second (so that one chunk's launches overtake the other's instead of sharing)
// A CU-mask bit i stands for CU i / 8 of XCD i % 8 (the driver deals the mask out round-robin over the XCCs).
static int32_t ensure_stream_layout(dapol_ctx* c, const char* name) {
    if (c->layout_name == name) return DAPOL_OK;
    for (int i = 0; i < 2; i++) if (c->layout_lane[i]) { (void)hipStreamDestroy(c->layout_lane[i]); c->layout_lane[i] = nullptr; }
    if (c->layout_msm) { (void)hipStreamDestroy(c->layout_msm); c->layout_msm = nullptr; }
    c->layout_name.clear();
    c->layout_split_msm = false;
    for (int i = 0; i < 2; i++) if (!c->layout_ev[i]) HIPCHK(hipEventCreateWithFlags(&c->layout_ev[i], hipEventDisableTiming));
    const int ncu = c->n_cu, words = (ncu + 31) / 32;
    auto mask_of = [&](auto pred) { std::vector<uint32_t> m((size_t)words, 0u); for (int i = 0; i < ncu; i++) if (pred(i)) m[i >> 5] |= 1u << (i & 31); return m; };
    const std::string n(name);
    if (n == "split_xcd" || n == "split_cu") {
        for (int h = 0; h < 2; h++) {
            auto m = n == "split_xcd" ? mask_of([&](int i) { return ((i % 8) < 4) == (h == 0); }) : mask_of([&](int i) { return (i < ncu / 2) == (h == 0); });
            HIPCHK(hipExtStreamCreateWithCUMask(&c->layout_lane[h], (uint32_t)words, m.data()));
        }
    } else if (n.rfind("msm:", 0) == 0) {
        const int K = atoi(n.c_str() + 4);
        if (K < 8 || K > ncu / 2 || K % 8) return fail(DAPOL_ERR_INVALID_ARGUMENT, "DAPOL_STREAM_LAYOUT=msm:<K>: K must be a multiple of 8 in [8, CUs / 2]");
        auto small = mask_of([&](int i) { return i < K; }), big = mask_of([&](int i) { return i >= K; });
        for (int h = 0; h < 2; h++) HIPCHK(hipExtStreamCreateWithCUMask(&c->layout_lane[h], (uint32_t)words, small.data()));
        HIPCHK(hipExtStreamCreateWithCUMask(&c->layout_msm, (uint32_t)words, big.data()));
        c->layout_split_msm = true;
    } else if (n == "prio" || n == "prio_lanes") {
        int lo = 0, hi = 0;                                   // (numerically lower = higher priority)
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        if (n == "prio") {
            for (int h = 0; h < 2; h++) HIPCHK(hipStreamCreateWithPriority(&c->layout_lane[h], hipStreamDefault, hi));
            HIPCHK(hipStreamCreateWithPriority(&c->layout_msm, hipStreamDefault, lo));
            c->layout_split_msm = true;
        } else {
            HIPCHK(hipStreamCreateWithPriority(&c->layout_lane[0], hipStreamDefault, lo));
            HIPCHK(hipStreamCreateWithPriority(&c->layout_lane[1], hipStreamDefault, hi));
        }
    } else return fail(DAPOL_ERR_INVALID_ARGUMENT, "unknown DAPOL_STREAM_LAYOUT");
    c->layout_name = n;
    return DAPOL_OK;
}

// A FORK hands kernels that read and write the context's scratch (or a call's own buffers) to a side stream; the matching JOIN makes
// the context's stream wait for them.  An early return between the two -- any HIPCHK / LAUNCH_CHECK -- would leave those kernels
// running while the caller's next call reuses the scratch, or after the call's buffers are freed.  This guard, one per forking
// function, waits for every side stream that was forked and not yet joined when the function is left.  (Nothing to wait for on the
// normal path: the join has closed it.)
static int32_t ctx_make_streams(dapol_ctx* c) {
    HIPCHK(hipHostMalloc((void**)&c->h_pinned, 64, hipHostMallocDefault));
    HIPCHK(hipStreamCreate(&c->stream));
    HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    for (int i = 0; i < 3; i++) {
        HIPCHK(hipStreamCreate(&c->side[i]));
        HIPCHK(hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming));
    }
    HIPCHK(hipStreamCreate(&c->msm_stream));
    for (int i = 0; i < 4; i++) HIPCHK(hipEventCreateWithFlags(&c->ev_v[i], hipEventDisableTiming));
    for (int i = 0; i < 4; i++) {
        HIPCHK(hipEventCreateWithFlags(&c->ev_msm_pre[i], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->ev_msm_post[i], hipEventDisableTiming));
    }
    return DAPOL_OK;
}
static void ctx_free_streams(dapol_ctx* ctx) {
    if (ctx->h_pinned) { (void)hipHostFree(ctx->h_pinned); ctx->h_pinned = nullptr; }
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    for (int i = 0; i < 3; i++) {
        if (ctx->side[i]) (void)hipStreamDestroy(ctx->side[i]);
        if (ctx->ev_join[i]) (void)hipEventDestroy(ctx->ev_join[i]);
    }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    for (int i = 0; i < 4; i++) if (ctx->ev_v[i]) (void)hipEventDestroy(ctx->ev_v[i]);
    if (ctx->msm_stream) (void)hipStreamDestroy(ctx->msm_stream);
    for (int i = 0; i < 2; i++) {
        if (ctx->layout_lane[i]) (void)hipStreamDestroy(ctx->layout_lane[i]);
        if (ctx->layout_ev[i]) (void)hipEventDestroy(ctx->layout_ev[i]);
    }
    if (ctx->layout_msm) (void)hipStreamDestroy(ctx->layout_msm);
    for (int i = 0; i < 4; i++) {
        if (ctx->ev_msm_pre[i]) (void)hipEventDestroy(ctx->ev_msm_pre[i]);
        if (ctx->ev_msm_post[i]) (void)hipEventDestroy(ctx->ev_msm_post[i]);
    }
}
// Lane i (0 = the context itself) for one group of a small call's sub-proofs: see dapol_ctx::aux.
static int32_t ctx_lane(dapol_ctx* ctx, int i, dapol_ctx** out) {
    if (i == 0) { *out = ctx; return DAPOL_OK; }
    dapol_ctx*& a = ctx->aux[i - 1];
    if (!a) {
        dapol_ctx* n = new dapol_ctx();
        n->opt = ctx->opt; n->device = ctx->device; n->max_parties = ctx->max_parties; n->n_cu = ctx->n_cu;
        n->msm_waves_per_cu = ctx->msm_waves_per_cu; n->msm_dyn_lds = ctx->msm_dyn_lds;
        n->tv = ctx->tv;                                  // the same tables: only the primary owns (and frees) them
        int32_t rc = ctx_make_streams(n);
        if (rc) { ctx_free_streams(n); delete n; return rc; }
        a = n;
    }
    a->opt = ctx->opt;                                    // (dapol_ctx_set_options may have changed since the lane was made)
    *out = a;
    return DAPOL_OK;
}

struct ForkGuard {
    dapol_ctx* c;
    bool open[3] = {false, false, false};
    hipStream_t extra = nullptr;     // one more stream that carries this call's kernels while the guard is armed (the measurement knobs'
    bool extra_open = false;         // MSM stream: DAPOL_MSM_SERIAL / stream layouts); closed by done() on the normal path
    explicit ForkGuard(dapol_ctx* c_) : c(c_) {}
    ForkGuard(const ForkGuard&) = delete;
    ForkGuard& operator=(const ForkGuard&) = delete;
    void forked(int i) { open[i] = true; }
    void joined(int i) { open[i] = false; }
    void also(hipStream_t s) { extra = s; extra_open = s != nullptr; }
    void done() { extra_open = false; }
    ~ForkGuard() {
        for (int i = 0; i < 3; i++)
            if (open[i]) { (void)hipStreamSynchronize(c->side[i]); g_fork_guard_waits.fetch_add(1, std::memory_order_relaxed); }
        if (extra_open) { (void)hipStreamSynchronize(extra); g_fork_guard_waits.fetch_add(1, std::memory_order_relaxed); }
    }
};
// Test knob (behind BOTH opt-ins, test_knob above): DAPOL_TEST_FAIL_AFTER_FORK=<site> makes the named site return an error right
// after its fork, as a failed launch would -- tests/test_gpu_fault_paths.py then checks that the next call on the context is clean.
#define FAULT_AFTER_FORK(site)                                                                                          \
    do {                                                                                                                \
        const char* e_ = test_knob("DAPOL_TEST_FAIL_AFTER_FORK");                                                         \
        if (e_ && !strcmp(e_, site)) return fail(DAPOL_ERR_HIP, "injected failure after the fork at " site " (test knob)"); \
    } while (0)

// Combined (random-linear-combination) verification checks that FAILED and went on to bisection / the proof-by-proof check.  On
// batches of valid proofs this stays 0; a test that verifies two different all-valid batches back to back asserts exactly that
// (a stale-scratch dependence between passes shows up as a spurious failure here long before it shows up in a verdict).
static std::atomic<unsigned long long> g_verify_fallbacks{0};
int32_t dapol_diag_verify_fallbacks(uint64_t* count) {
    if (!count) return fail(DAPOL_ERR_INVALID_ARGUMENT, "null argument");
    *count = g_verify_fallbacks.load(std::memory_order_relaxed);
    return DAPOL_OK;
}
// Device time of the proving pipeline of the last dapol_range_prove_batch (HIP events on the context's stream around
// range_prove_device: inputs already in HBM, proofs not yet copied back) -- what tools/bench_small_parties.py quotes.
static std::atomic<double> g_last_range_prove_ms{0.0};
int32_t dapol_diag_range_prove_ms(double* ms) {
    if (!ms) return fail(DAPOL_ERR_INVALID_ARGUMENT, "null argument");
    *ms = g_last_range_prove_ms.load(std::memory_order_relaxed);
    return DAPOL_OK;
}
int32_t dapol_diag_fork_guard_waits(uint64_t* count) {
    if (!count) return fail(DAPOL_ERR_INVALID_ARGUMENT, "null argument");
    *count = g_fork_guard_waits.load(std::memory_order_relaxed);
    return DAPOL_OK;
}

int32_t dapol_ctx_create(int32_t device, int32_t max_parties, int32_t digest_id, dapol_ctx** out) {
    return dapol_ctx_create_opts(device, max_parties, digest_id, nullptr, out);
}
int32_t dapol_ctx_get_options(dapol_ctx* ctx, dapol_options* out) {
    if (!ctx || !out) return fail(DAPOL_ERR_INVALID_ARGUMENT, "null argument");
    *out = ctx->opt;
    out->struct_size = (int32_t)sizeof(dapol_options);
    out->window_bits = ctx->tv.wbits;
    out->high_half_rows = ctx->tv.hi_split ? 1 : -1;
    return DAPOL_OK;
}
int32_t dapol_ctx_set_options(dapol_ctx* ctx, const dapol_options* o) {
    if (!ctx || !o) return fail(DAPOL_ERR_INVALID_ARGUMENT, "null argument");
    if (!options_ok(o)) return fail(DAPOL_ERR_INVALID_ARGUMENT, "bad dapol_options (struct_size, or a field out of range)");
    const dapol_options keep = ctx->opt;
    ctx->opt = *o;
    ctx->opt.window_bits = keep.window_bits; ctx->opt.table_gb = keep.table_gb; ctx->opt.high_half_rows = keep.high_half_rows;    // fixed at creation
    ctx->opt.profile = keep.profile;
    return DAPOL_OK;
}
int32_t dapol_ctx_create_opts(int32_t device, int32_t max_parties, int32_t digest_id, const dapol_options* options, dapol_ctx** out) {
    if (!out) return fail(DAPOL_ERR_INVALID_ARGUMENT, "out is null");
    *out = nullptr;
    if (!options_ok(options)) return fail(DAPOL_ERR_INVALID_ARGUMENT, "bad dapol_options (struct_size, or a field out of range)");
    if (digest_id != DAPOL_DIGEST_BLAKE3 && digest_id != DAPOL_DIGEST_BLAKE2S && digest_id != DAPOL_DIGEST_BLAKE2B)
        return fail(DAPOL_ERR_INVALID_DIGEST_SIZE, "node digest must be BLAKE3, Blake2s-256 or Blake2b-512");
    if (max_parties < 1 || max_parties > 1024 || (max_parties & (max_parties - 1)))
        return fail(DAPOL_ERR_INVALID_ARGUMENT, "max_parties must be a power of two in [1, 1024]");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count)
        return fail(DAPOL_ERR_NO_DEVICE, "no usable HIP device");
    HIPCHK(hipSetDevice(device));
    dapol_ctx* c = new dapol_ctx();
    if (options) c->opt = *options;
    c->device = device;
    c->max_parties = max_parties;
    {
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, device));
        if (prop.multiProcessorCount > 0) c->n_cu = prop.multiProcessorCount;
        // Residency of the dominant kernel (one wavefront per block): what its registers and LDS allow -- asked of the runtime,
        // not assumed.  DAPOL_MSM_OCC_CAP=<waves per SIMD> lowers it by padding the launch's LDS (A/B knob: at the socket power
        // cap more resident wavefronts are not automatically faster, profiles/r01_madchain_ab.txt).
        if (const char* e = knob("DAPOL_MSM_OCC_CAP")) {
            int cap = atoi(e);
            const size_t lds_cu = prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor : 163840, stat = 0;   // (the kernel itself uses no LDS)
            if (cap >= 1 && cap <= 8) {
                size_t per_block = lds_cu / (size_t)(4 * cap) / 512 * 512;          // LDS is granted in 512-byte granules
                if (per_block > stat) c->msm_dyn_lds = (unsigned)(per_block - stat);
            }
        }
        int blocks = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_rp_msm<MSM_PLAIN, 4>, 64, c->msm_dyn_lds) == hipSuccess && blocks > 0)
            c->msm_waves_per_cu = blocks;
        else (void)hipGetLastError();
    }
    struct Guard { dapol_ctx* c; ~Guard() { if (c) dapol_ctx_destroy(c); } } guard{c};
    { int32_t rc_ = ctx_make_streams(c); if (rc_) return rc_; }
    const int P = max_parties;
    // window width: the widest (<= 17 bits: wider measured slower, profiles/r01_wbits_ab4.txt) whose tables fit the budget --
    // DAPOL_TABLE_GB if set, else 40 GB but never more than 30 % of the memory that is free right now (a second context on
    // the same GPU, or a smaller device, gets narrower windows instead of an allocation failure) -- or DAPOL_WBITS (up to 20)
    int wbits = WBITS_MIN;
    {
        const char* eb = knob("DAPOL_TABLE_GB");
        double budget = 40.0e9;
        size_t free_b = 0, total_b = 0;
        if (c->opt.profile == DAPOL_PROFILE_HOST) budget = 18.0e9;       // 16-bit windows for 32 parties (17.3 GB), 15-bit for 64
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (double)free_b * 0.30 < budget) budget = (double)free_b * 0.30;
        if (c->opt.table_gb > 0) budget = c->opt.table_gb * 1e9;
        if (eb) budget = atof(eb) * 1e9;
        for (int w = WBITS_MIN; w <= WBITS_AUTO_MAX; w++) {
            TableView t{nullptr, P, w, 0, 0};
            if ((double)t.n_rows() * (double)t.row_words() * 4.0 <= budget) wbits = w;
        }
        const char* ew = knob("DAPOL_WBITS");
        if (c->opt.window_bits) wbits = c->opt.window_bits;
        if (ew) {
            int w = atoi(ew);
            if (w < WBITS_MIN || w > WBITS_MAX) return fail(DAPOL_ERR_INVALID_ARGUMENT, "DAPOL_WBITS must be in [8, 20]");
            wbits = w;
        }
    }
    TableView tv{nullptr, P, wbits, digest_id, 0};
    const int rows = tv.n_rows();
    {   // High-half rows for the G / H generators (tables.h): worth their memory (as much again as the G / H rows) only for the
        // prover's materialisation step, so only when they fit beside everything else: DAPOL_TABLE_HI=0 / 1 forces, default = on
        // when the doubled tables stay below 30 % of the free memory and the context is a prover's (<= 64 parties).  Measured
        // interleaved (profiles/r02_hi_rows_ab.txt): +0.7 % at 2^18 proofs, +1.5 % at 2^20, for 34 GB more of HBM.
        size_t free_b = 0, total_b = 0;
        const double bytes2 = (double)(rows + 128 * P) * (double)tv.row_words() * 4.0;
        bool hi = P <= 64 && hipMemGetInfo(&free_b, &total_b) == hipSuccess && bytes2 <= 0.30 * (double)free_b;
        if (c->opt.profile == DAPOL_PROFILE_HOST) hi = false;             // as much memory again for +1.5-2 %: not for a shared GPU
        if (c->opt.high_half_rows) hi = c->opt.high_half_rows > 0 && P <= 64;
        if (const char* e = knob("DAPOL_TABLE_HI")) hi = atoi(e) != 0;
        if (hi) tv.hi_split = (tv.nwin_c() + 1) / 2;
    }
    const int rows_total = tv.n_rows_total();
    DevBuf<uint32_t> uniform;
    DevBuf<int32_t> base_pts;
    HIPCHK(uniform.alloc((size_t)2 * P * 64 * 16));
    HIPCHK(base_pts.alloc((size_t)rows_total * 40));
    HIPCHK(c->table.alloc((size_t)rows_total * tv.row_words()));
    HIPCHK(c->gens_comp.alloc((size_t)rows * 8));
    hipLaunchKernelGGL(k_ctx_chains, dim3(nblk(2 * P, 64)), dim3(64), 0, c->stream, uniform.p, P);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_ctx_points, dim3(nblk(128 * P, 64)), dim3(64), 0, c->stream, base_pts.p, uniform.p, P);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_ctx_pedersen, dim3(1), dim3(64), 0, c->stream, base_pts.p, P, wbits, tv.nwin());
    LAUNCH_CHECK();
    if (tv.hi_split) {
        hipLaunchKernelGGL(k_ctx_hi_points, dim3(nblk(128 * P, 64)), dim3(64), 0, c->stream, base_pts.p, 128 * P, rows, wbits * tv.hi_split);
        LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_ctx_table, dim3(nblk((size_t)rows_total * tv.entries(), 64)), dim3(64), 0, c->stream, c->table.p, base_pts.p, rows_total, wbits,
                       tv.entries());
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_ctx_compress, dim3(nblk(rows, 64)), dim3(64), 0, c->stream, c->gens_comp.p, base_pts.p, rows);
    LAUNCH_CHECK();
    HIPCHK(hipStreamSynchronize(c->stream));
    c->tv = tv;
    c->tv.base = c->table.p;
    guard.c = nullptr;
    *out = c;
    return DAPOL_OK;
}

static void ctx_retain(dapol_ctx* ctx) { ctx->refs.fetch_add(1); }
int32_t dapol_ctx_destroy(dapol_ctx* ctx) {
    if (!ctx) return DAPOL_OK;
    if (ctx->refs.fetch_sub(1) > 1) return DAPOL_OK;          // trees / workloads still use it: the last of them frees it
    (void)hipSetDevice(ctx->device);
    for (int i = 0; i < 3; i++)
        if (ctx->aux[i]) {
            ctx->aux[i]->scratch.release();
            ctx->aux[i]->vio.release();
            ctx_free_streams(ctx->aux[i]);
            delete ctx->aux[i];
            ctx->aux[i] = nullptr;
        }
    ctx->scratch.release();
    ctx->vio.release();
    ctx->table.release();
    ctx->gens_comp.release();
    ctx_free_streams(ctx);
    delete ctx;
    return DAPOL_OK;
}

int32_t dapol_ctx_generator(dapol_ctx* ctx, int32_t which, int32_t party, int32_t bit, uint8_t out32[32]) {
    if (!ctx || !out32) return fail(DAPOL_ERR_INVALID_ARGUMENT, "null argument");
    int row;
    if (which == 0) row = ctx->tv.row_B(0);
    else if (which == 1) row = ctx->tv.row_Bb(0);
    else if ((which == 2 || which == 3) && party >= 0 && party < ctx->max_parties && bit >= 0 && bit < 64)
        row = which == 2 ? ctx->tv.row_G(party, bit) : ctx->tv.row_H(party, bit);
    else return fail(DAPOL_ERR_INVALID_ARGUMENT, "bad generator selector");
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpy(out32, ctx->gens_comp.p + (size_t)row * 8, 32, hipMemcpyDeviceToHost));
    return DAPOL_OK;
}

// ------------------------------------------------------------------------------------------- commitments
int32_t dapol_commit_hash_batch(dapol_ctx* ctx, size_t n, const uint64_t* v, const uint8_t* r32, uint8_t* C_out32, uint8_t* H_out32) {
    if (!ctx || (n && (!v || !r32 || !C_out32 || !H_out32))) return fail(DAPOL_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return DAPOL_OK;
    HIPCHK(hipSetDevice(ctx->device));
    DevBuf<uint64_t> dv;
    DevBuf<uint32_t> dr, dC, dH, dHw;
    HIPCHK(dv.alloc(n)); HIPCHK(dr.alloc(n * 8)); HIPCHK(dC.alloc(n * 8)); HIPCHK(dH.alloc(n * 8));
    HIPCHK(hipMemcpyAsync(dv.p, v, n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(dr.p, r32, n * 32, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_commit_hash, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, ctx->tv, n, dv.p, dr.p, dC.p, dH.p, (int32_t*)nullptr);
    LAUNCH_CHECK();
    if (ctx_wide(ctx)) {                           // 64-byte digest: H = D(C) over the commitments just made
        HIPCHK(dHw.alloc(n * 16));
        hipLaunchKernelGGL(k_wide_hash_leaves, dim3(nblk(n, 256)), dim3(256), 0, ctx->stream, n, dC.p, dHw.p);
        LAUNCH_CHECK();
    }
    HIPCHK(hipMemcpyAsync(C_out32, dC.p, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(H_out32, ctx_wide(ctx) ? dHw.p : dH.p, n * ctx_hash_bytes(ctx), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return DAPOL_OK;
}

// --------------------------------------------------------------------------------------------------- tree
template <typename T>
struct Span { T* p = nullptr; };             // a view into the tree's arena (same `.p` spelling as DevBuf)
struct LevelBuf {
    size_t n = 0;                            // real nodes of the level (the host-side bound while the build is in flight)
    Span<uint64_t> idx, v;
    Span<uint32_t> C, H, r, padC, padH, padr, parent;
    Span<uint8_t> has_pad;
};

struct dapol_tree {
    dapol_ctx* ctx = nullptr;
    int height = 0;
    std::vector<LevelBuf> levels;      // 0 = leaves .. height = root
    DevBuf<uint8_t> arena;             // every level's arrays: ONE allocation per build
    // level 0 may borrow caller-resident device arrays (workload path)
    uint64_t* leaf_idx = nullptr;
    uint64_t* leaf_v = nullptr;
    uint32_t* leaf_r = nullptr;
    uint64_t n_pad = 0, n_real = 0;
    int index_bits = 0, shard_bits = 0;          // what the tree was built with (dapol_tree_update rebuilds with the same)
    uint8_t pad_seed[32] = {0};
    bool invalid = false;              // an in-place update failed after its first write: root and leaves may disagree; every call refuses the tree
    bool tape_built = false;           // dapol_tree_build_tape: the padding draws came from a caller's tape (no seed to make further ones from)
    DevBuf<LevelView> d_views;         // view(0..height, nullptr) on the device, for kernels that walk several levels
    // 64-byte node hashes (a Blake2b context): the hash chain laid over the built tree (tree_hash_wide), per level H16[n] | padH16[n]
    DevBuf<uint32_t> wide;
    std::vector<WideView> wviews;      // host copy: pointers into `wide`
    DevBuf<WideView> d_wviews;
    LevelView view(int k, int32_t* ext) {
        LevelBuf& L = levels[k];
        LevelView lv;
        lv.n = L.n;
        lv.idx = k == 0 ? leaf_idx : L.idx.p;
        lv.v = k == 0 ? leaf_v : L.v.p;
        lv.r = k == 0 ? leaf_r : L.r.p;
        lv.C = L.C.p; lv.H = L.H.p; lv.padC = L.padC.p; lv.padH = L.padH.p; lv.padr = L.padr.p;
        lv.has_pad = L.has_pad.p; lv.parent = L.parent.p; lv.ext = ext;
        return lv;
    }
};

// dapol_tree_update re-merges in place; an error between its first write and its last (a HIP failure) leaves a tree whose upper
// levels no longer match its leaves.  Such a tree is marked and every entry point refuses it, loudly, instead of proving from it.
static int32_t tree_usable(const dapol_tree* t) {
    if (t && t->invalid)
        return fail(DAPOL_ERR_INVALID_ARGUMENT, "the tree was left inconsistent by an in-place update that failed midway: destroy it and build it again");
    return DAPOL_OK;
}
#define TREE_USABLE(t) do { int32_t rc_ = tree_usable(t); if (rc_) return rc_; } while (0)
struct TreePoison {                    // armed before the first write of an in-place update, disarmed when the last one has completed
    dapol_tree* t;
    bool armed = false;
    ~TreePoison() { if (armed) t->invalid = true; }
};

// Builds the tree from device-resident leaf arrays (d_idx sorted; d_r is masked in place).  own==true: the tree
// takes ownership of nothing; leaf arrays must outlive it (they are owned by the caller-side holder below).
static int32_t tree_build_device_core(dapol_ctx* ctx, int index_bits, int shard_bits, size_t n, uint64_t* d_idx, uint64_t* d_v, uint32_t* d_r,
                                      const uint8_t pad_seed32[32], dapol_tree* t, const uint32_t* d_tape = nullptr, size_t tape_draws = 0) {
    hipStream_t st = ctx->stream;
    t->tape_built = d_tape != nullptr;
    const int height = index_bits - shard_bits;       // levels built on this GPU
    t->ctx = ctx;
    t->height = height;
    t->leaf_idx = d_idx; t->leaf_v = d_v; t->leaf_r = d_r;
    t->index_bits = index_bits; t->shard_bits = shard_bits;
    memcpy(t->pad_seed, pad_seed32, 32);
    t->levels.clear();
    t->levels.resize((size_t)height + 1);
    t->n_pad = 0;
    t->n_real = 0;
    DevBuf<uint32_t> bad, seed, cnt;
    HIPCHK(bad.alloc(1)); HIPCHK(seed.alloc(8)); HIPCHK(cnt.alloc((size_t)height + 2));
    HIPCHK(hipMemsetAsync(bad.p, 0, 4, st));
    HIPCHK(hipMemcpyAsync(seed.p, pad_seed32, 32, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_tree_check_leaves, dim3(nblk(n, 256)), dim3(256), 0, st, n, d_idx, index_bits, height, bad.p);
    LAUNCH_CHECK();
    uint32_t h_bad = 0;
    HIPCHK(hipMemcpyAsync(&h_bad, bad.p, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));          // the one early wait: malformed input must not reach the level kernels
    if (h_bad) return fail(DAPOL_ERR_INVALID_ARGUMENT, "leaf indexes must be strictly increasing, below 2^height, and (shard build) share their top shard_bits bits");
    // Upper bound of every level's size, known on the host: a level has at most as many nodes as the one below and at most
    // 2^(levels above it) positions.  The actual sizes are computed on the device (cnt) and read back once, at the end.
    std::vector<size_t> bound((size_t)height + 1);
    bound[0] = n;
    for (int k = 0; k < height; k++) {
        const int bits_above = height - (k + 1);
        bound[k + 1] = bound[k];
        if (bits_above < 40 && ((size_t)1 << bits_above) < bound[k + 1]) bound[k + 1] = (size_t)1 << bits_above;
    }
    {   // one arena for all levels
        size_t need = 0, pad_total = 0;
        auto take = [&](size_t bytes) { size_t o = need; need += align_up(bytes, 256); return o; };
        std::vector<size_t> off((size_t)(height + 1) * 10);
        for (int k = 0; k <= height; k++) {
            const size_t c = bound[k];
            size_t* o = &off[(size_t)k * 10];
            o[0] = k ? take(c * 8) : 0; o[1] = k ? take(c * 8) : 0; o[2] = k ? take(c * 32) : 0;      // idx, v, r (level 0 borrows the caller's)
            o[3] = take(c * 32); o[4] = take(c * 32); o[5] = take(c * 32); o[6] = take(c * 32); o[7] = take(c * 32); o[8] = take(c * 4);
            pad_total += align_up(c, 256);
        }
        const size_t pad_at = need;
        need += pad_total;
        if (t->arena.n < need) HIPCHK(t->arena.alloc(need));      // (a workload hands the arena of its previous build on: no hipFree / hipMalloc per step)
        HIPCHK(hipMemsetAsync(t->arena.p + pad_at, 0, pad_total, st));
        size_t pad_off = pad_at;
        for (int k = 0; k <= height; k++) {
            LevelBuf& L = t->levels[k];
            const size_t* o = &off[(size_t)k * 10];
            uint8_t* a = t->arena.p;
            L.n = bound[k];
            if (k) { L.idx.p = (uint64_t*)(a + o[0]); L.v.p = (uint64_t*)(a + o[1]); L.r.p = (uint32_t*)(a + o[2]); }
            L.C.p = (uint32_t*)(a + o[3]); L.H.p = (uint32_t*)(a + o[4]); L.padC.p = (uint32_t*)(a + o[5]); L.padH.p = (uint32_t*)(a + o[6]);
            L.padr.p = (uint32_t*)(a + o[7]); L.parent.p = (uint32_t*)(a + o[8]);
            L.has_pad.p = a + pad_off;
            pad_off += align_up(bound[k], 256);
        }
    }
    const uint32_t n32 = (uint32_t)n;
    HIPCHK(hipMemcpyAsync(cnt.p, &n32, 4, hipMemcpyHostToDevice, st));
    const bool phased = n <= (size_t)TREE_SMALL_MAX && height >= 1 && !knob("DAPOL_TREE_LEVELWISE") && !d_tape;     // (tape mode: the level-wise kernel reads the tape)
    DevBuf<uint32_t> tape_short;
    HIPCHK(tape_short.alloc(1));
    HIPCHK(hipMemsetAsync(tape_short.p, 0, 4, st));
    const PadTape ptape{d_tape, (uint32_t)std::min<size_t>(tape_draws, 0xffffffffu), tape_short.p};
    if (phased) {
        // Small trees, by phases (kernels_ctx_tree.h, "small trees"): structure, all padding nodes, point sums level by level, all
        // encodings, hashes level by level.  The extended points of every level are kept until the encodings are done.
        std::vector<uint32_t> h_off((size_t)height + 2);
        size_t tot = 0;
        for (int k = 0; k <= height; k++) { h_off[k] = (uint32_t)tot; tot += bound[k]; }
        h_off[(size_t)height + 1] = (uint32_t)tot;
        std::vector<LevelView> hv((size_t)height + 1);
        for (int k = 0; k <= height; k++) hv[k] = t->view(k, nullptr);
        // temporaries from the context's scratch (a fresh hipMalloc of a few MB costs more than the whole build)
        const size_t b_ext = align_up(tot * 160, 256), b_pad = align_up((size_t)h_off[height] * 160, 256), b_off = align_up(h_off.size() * 4, 256);
        HIPCHK(ctx->scratch.ensure(b_ext + b_pad + b_off));
        int32_t* const ext_all = (int32_t*)ctx->scratch.p;
        int32_t* const extpad_all = (int32_t*)((uint8_t*)ctx->scratch.p + b_ext);
        uint32_t* const d_off = (uint32_t*)((uint8_t*)ctx->scratch.p + b_ext + b_pad);
        if (t->d_views.n != hv.size()) HIPCHK(t->d_views.alloc(hv.size()));
        HIPCHK(hipMemcpyAsync(t->d_views.p, hv.data(), hv.size() * sizeof(LevelView), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(d_off, h_off.data(), h_off.size() * 4, hipMemcpyHostToDevice, st));
        // the leaves' commitments do not depend on the structure: they run on a side stream beside S and P
        ForkGuard fg(ctx);
        HIPCHK(hipEventRecord(ctx->ev_fork, st));
        HIPCHK(hipStreamWaitEvent(ctx->side[0], ctx->ev_fork, 0));
        fg.forked(0);
        hipLaunchKernelGGL(k_commit_hash, dim3(nblk(n, 64)), dim3(64), 0, ctx->side[0], ctx->tv, n, d_v, d_r, t->levels[0].C.p, t->levels[0].H.p, ext_all);
        LAUNCH_CHECK();
        HIPCHK(hipEventRecord(ctx->ev_join[0], ctx->side[0]));
        FAULT_AFTER_FORK("tree");
        hipLaunchKernelGGL(k_tree_structure_small, dim3(1), dim3(1024), 0, st, height, t->d_views.p, cnt.p);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(k_tree_padding_all, dim3(nblk(h_off[height], 64)), dim3(64), 0, st, ctx->tv, t->d_views.p, height, cnt.p, d_off, seed.p, extpad_all);
        LAUNCH_CHECK();
        HIPCHK(hipStreamWaitEvent(st, ctx->ev_join[0], 0));
        fg.joined(0);
        for (int k = 0; k < height; k++) {
            hipLaunchKernelGGL(k_tree_sum_level, dim3(nblk(bound[k], 64)), dim3(64), 0, st, hv[k], hv[k + 1], k, cnt.p, ext_all + (size_t)h_off[k] * 40,
                               extpad_all + (size_t)h_off[k] * 40, ext_all + (size_t)h_off[k + 1] * 40);
            LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(k_tree_compress_all, dim3(nblk(tot - h_off[1], 64)), dim3(64), 0, st, t->d_views.p, height, cnt.p, d_off, ext_all);
        LAUNCH_CHECK();
        for (int k = 0; k < height; k++) {
            hipLaunchKernelGGL(k_tree_hash_level, dim3(nblk(bound[k], 64)), dim3(64), 0, st, ctx->tv.digest, hv[k], hv[k + 1], k, cnt.p);
            LAUNCH_CHECK();
        }
    } else {
    // The padding children of a level are made in a launch of their own (k_tree_pad_level) and the merge reads them back
    // (k_tree_merge<1>): fused, the kernel needed 262 VGPRs + 6 AGPRs and 624 bytes of scratch per lane -- one wavefront per SIMD --;
    // apart, 178 and 226 VGPRs, two wavefronts each: 2^20 leaves x height 32 in 36.0 ms instead of 43.8, same root
    // (profiles/r6_tree_split_ab.txt).  Tape mode keeps the fused kernel (it reads the tape by rank); DAPOL_TREE_SPLIT=0 restores it.
    const bool split_pad = !d_tape && !(knob("DAPOL_TREE_SPLIT") && atoi(knob("DAPOL_TREE_SPLIT")) == 0);
    // temporaries out of the context's scratch (as the phased path): seven hipMalloc / hipFree pairs per build otherwise, and a
    // hipFree waits for the device
    struct { uint32_t *flag, *pos, *head, *bsums; int32_t *ext_a, *ext_b, *ext_pad; } tmp;
    {
        size_t need = 0;
        auto take = [&](size_t bytes) { size_t o = need; need += align_up(bytes, 256); return o; };
        const size_t o_flag = take(n * 4), o_pos = take(n * 4), o_head = take(n * 4), o_bs = take((nblk(n, 1024) + 1) * 4), o_a = take(n * 160), o_b = take(n * 160),
                     o_p = take(split_pad ? n * 160 : 16);
        HIPCHK(ctx->scratch.ensure(need));
        uint8_t* b = (uint8_t*)ctx->scratch.p;
        tmp.flag = (uint32_t*)(b + o_flag); tmp.pos = (uint32_t*)(b + o_pos); tmp.head = (uint32_t*)(b + o_head); tmp.bsums = (uint32_t*)(b + o_bs);
        tmp.ext_a = (int32_t*)(b + o_a); tmp.ext_b = (int32_t*)(b + o_b); tmp.ext_pad = (int32_t*)(b + o_p);
    }
    hipLaunchKernelGGL(k_commit_hash, dim3(nblk(n, 256)), dim3(256), 0, st, ctx->tv, n, d_v, d_r, t->levels[0].C.p, t->levels[0].H.p, tmp.ext_a);
    LAUNCH_CHECK();
    int32_t* ext_cur = tmp.ext_a;
    int32_t* ext_nxt = tmp.ext_b;
    for (int k = 0; k < height; k++) {                    // launches only: nothing here waits for the device
        LevelView cur = t->view(k, ext_cur);
        hipLaunchKernelGGL(k_tree_flags, dim3(nblk(bound[k], 256)), dim3(256), 0, st, cnt.p + k, cur.idx, tmp.flag);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(k_scan_block, dim3(nblk(bound[k], 1024)), dim3(256), 0, st, cnt.p + k, tmp.flag, tmp.pos, tmp.bsums);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, st, cnt.p + k, tmp.bsums, cnt.p + k + 1);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(k_scan_finish, dim3(nblk(bound[k], 256)), dim3(256), 0, st, cnt.p + k, tmp.flag, tmp.pos, tmp.bsums, tmp.head);
        LAUNCH_CHECK();
        LevelView nxt = t->view(k + 1, k + 1 < height ? ext_nxt : nullptr);
        if (split_pad) {
            hipLaunchKernelGGL(k_tree_pad_level, dim3(nblk(bound[k + 1], 256)), dim3(256), 0, st, ctx->tv, cur, tmp.head, k, seed.p, cnt.p, tmp.ext_pad);
            LAUNCH_CHECK();
            hipLaunchKernelGGL(k_tree_merge<1>, dim3(nblk(bound[k + 1], 256)), dim3(256), 0, st, ctx->tv, cur, nxt, tmp.head, k, seed.p, cnt.p, ptape, tmp.ext_pad);
        } else
            hipLaunchKernelGGL(k_tree_merge<0>, dim3(nblk(bound[k + 1], 256)), dim3(256), 0, st, ctx->tv, cur, nxt, tmp.head, k, seed.p, cnt.p, ptape, (const int32_t*)nullptr);
        LAUNCH_CHECK();
        std::swap(ext_cur, ext_nxt);
    }
    }
    std::vector<uint32_t> h_cnt((size_t)height + 1);
    uint32_t h_short = 0;
    HIPCHK(hipMemcpyAsync(h_cnt.data(), cnt.p, ((size_t)height + 1) * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&h_short, tape_short.p, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (h_short) return fail(DAPOL_ERR_INVALID_ARGUMENT, "the padding tape is shorter than the tree's padding nodes (dapol_tree_padding_positions gives the count)");
    for (int k = 0; k <= height; k++) {
        t->levels[k].n = h_cnt[k];
        t->n_real += h_cnt[k];
        if (k < height) t->n_pad += 2 * (uint64_t)h_cnt[k + 1] - h_cnt[k];
    }
    std::vector<LevelView> hv((size_t)height + 1);
    for (int k = 0; k <= height; k++) hv[k] = t->view(k, nullptr);
    if (t->d_views.n != hv.size()) HIPCHK(t->d_views.alloc(hv.size()));
    HIPCHK(hipMemcpy(t->d_views.p, hv.data(), hv.size() * sizeof(LevelView), hipMemcpyHostToDevice));
    return DAPOL_OK;
}

// The hash chain of a 64-byte digest over a tree whose structure, commitments and padding nodes are in place (after a build or an
// update): level 0 = D(C), then one launch per level (kernels_ctx_tree.h, "64-byte node hashes").  The whole chain is redone after
// every update -- a pass over the tree's nodes, milliseconds at the reference test's sizes; only the 32-byte digests have the
// incremental re-hash.
static int32_t tree_hash_wide(dapol_tree* t) {
    dapol_ctx* ctx = t->ctx;
    if (!ctx_wide(ctx)) return DAPOL_OK;
    hipStream_t st = ctx->stream;
    const int H = t->height;
    size_t words = 0;
    std::vector<size_t> off((size_t)H + 1);
    for (int k = 0; k <= H; k++) { off[k] = words; words += 2 * t->levels[k].n * 16; }
    if (t->wide.n < words) HIPCHK(t->wide.alloc(words + words / 8));
    t->wviews.resize((size_t)H + 1);
    for (int k = 0; k <= H; k++) t->wviews[k] = WideView{t->wide.p + off[k], t->wide.p + off[k] + t->levels[k].n * 16};
    if (t->d_wviews.n != t->wviews.size()) HIPCHK(t->d_wviews.alloc(t->wviews.size()));
    HIPCHK(hipMemcpyAsync(t->d_wviews.p, t->wviews.data(), t->wviews.size() * sizeof(WideView), hipMemcpyHostToDevice, st));
    const size_t n0 = t->levels[0].n;
    hipLaunchKernelGGL(k_wide_hash_leaves, dim3(nblk(n0, 256)), dim3(256), 0, st, n0, t->levels[0].C.p, t->wviews[0].H);
    LAUNCH_CHECK();
    for (int k = 0; k < H; k++) {
        LevelView cur = t->view(k, nullptr);
        hipLaunchKernelGGL(k_wide_hash_level, dim3(nblk(cur.n, 256)), dim3(256), 0, st, cur, t->wviews[k], t->wviews[k + 1]);
        LAUNCH_CHECK();
    }
    HIPCHK(hipStreamSynchronize(st));              // (wviews is copied from a host vector that may be resized by the next call)
    return DAPOL_OK;
}
static int32_t tree_build_device(dapol_ctx* ctx, int index_bits, int shard_bits, size_t n, uint64_t* d_idx, uint64_t* d_v, uint32_t* d_r,
                                 const uint8_t pad_seed32[32], dapol_tree* t, const uint32_t* d_tape = nullptr, size_t tape_draws = 0) {
    int32_t rc = tree_build_device_core(ctx, index_bits, shard_bits, n, d_idx, d_v, d_r, pad_seed32, t, d_tape, tape_draws);
    return rc ? rc : tree_hash_wide(t);
}

struct OwnedLeaves {
    DevBuf<uint64_t> idx, v;
    DevBuf<uint32_t> r;
};
struct dapol_tree_owned : dapol_tree {
    OwnedLeaves leaves;          // empty when level 0 borrows the caller's device arrays (workload trees)
    DevBuf<uint8_t> upd_scratch; // dapol_tree_update's incremental path (kept: an update must not pay for an allocation)
    // A level that gains nodes (incremental insert) is rewritten out of place into one of two buffers of its own (ping-pong; the
    // arena region it came from is simply left behind).  cur = which of the two holds the level now (-1: still in the arena).
    struct LevelAlt { DevBuf<uint8_t> buf[2]; size_t cap[2] = {0, 0}; int cur = -1; };
    std::vector<LevelAlt> alt;
    int last_update_path = 0;    // what the last dapol_tree_update did: 0 rebuild, 1 replaced in place, 2 inserted in place, 3 both
    bool holds_ctx = false;      // API-created trees keep their context alive (workload trees live inside a workload that does)
};

int32_t dapol_tree_build(dapol_ctx* ctx, int32_t height, size_t n, const uint64_t* leaf_idx, const uint64_t* v, const uint8_t* r32,
                         const uint8_t pad_seed32[32], int32_t enforce_sparsity, dapol_tree** out) {
    if (!ctx || !out || !pad_seed32 || (n && (!leaf_idx || !v || !r32))) return fail(DAPOL_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (height < 0 || height > 64) return fail(DAPOL_ERR_TREE_HEIGHT_TOO_BIG, "tree height must not exceed 64");
    if (enforce_sparsity && ctx_wide(ctx))           // Dapol::new (src/dapol/mod.rs:101-103); new_blank + build (enforce_sparsity = 0) has no such check
        return fail(DAPOL_ERR_INVALID_DIGEST_SIZE, "digest size must be 32 bytes (DapolError::InvalidDigestSize)");
    if (enforce_sparsity && height < 64 && ((double)n * 2.0 > (double)(1ull << height) ))
        return fail(DAPOL_ERR_SPARSITY_TOO_SMALL, "2^height < 2 * number of liabilities");
    if (n == 0) return fail(DAPOL_ERR_INVALID_ARGUMENT, "empty leaf set");
    if (n > ((size_t)1 << 31)) return fail(DAPOL_ERR_INVALID_ARGUMENT, "at most 2^31 leaves per GPU (32-bit node positions); memory is the practical bound");
    HIPCHK(hipSetDevice(ctx->device));
    dapol_tree_owned* t = new dapol_tree_owned();
    struct Guard { dapol_tree* t; ~Guard() { if (t) dapol_tree_destroy(t); } } guard{t};
    HIPCHK(t->leaves.idx.alloc(n)); HIPCHK(t->leaves.v.alloc(n)); HIPCHK(t->leaves.r.alloc(n * 8));
    HIPCHK(hipMemcpyAsync(t->leaves.idx.p, leaf_idx, n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(t->leaves.v.p, v, n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(t->leaves.r.p, r32, n * 32, hipMemcpyHostToDevice, ctx->stream));
    int32_t rc = tree_build_device(ctx, height, 0, n, t->leaves.idx.p, t->leaves.v.p, t->leaves.r.p, pad_seed32, t);
    if (rc != DAPOL_OK) return rc;
    guard.t = nullptr;
    t->holds_ctx = true;
    ctx_retain(ctx);
    *out = t;
    return DAPOL_OK;
}

// Padding nodes of the tree over the given (sorted, distinct) leaves, in TAPE order: level bottom-up, index ascending.  Pure index
// arithmetic (smtree's build restated: every real node's missing sibling is a padding node; a parent exists iff a child does).
int32_t dapol_tree_padding_positions(int32_t height, size_t n, const uint64_t* leaf_idx, size_t* count, uint8_t* level, uint64_t* index) {
    if (!count || (n && !leaf_idx)) return fail(DAPOL_ERR_INVALID_ARGUMENT, "null argument");
    if (height < 0 || height > 64) return fail(DAPOL_ERR_TREE_HEIGHT_TOO_BIG, "tree height must not exceed 64");
    for (size_t i = 0; i < n; i++)
        if ((height < 64 && (leaf_idx[i] >> height)) || (i && leaf_idx[i] <= leaf_idx[i - 1]))
            return fail(DAPOL_ERR_INVALID_ARGUMENT, "leaf indexes must be strictly increasing and below 2^height");
    std::vector<uint64_t> cur(leaf_idx, leaf_idx + n), nxt;
    size_t k = 0;
    for (int L = 0; L < height; L++) {
        nxt.clear();
        for (size_t i = 0; i < cur.size();) {
            const uint64_t x = cur[i];
            if (!(x & 1) && i + 1 < cur.size() && cur[i + 1] == x + 1) i += 2;
            else {
                if (level) level[k] = (uint8_t)L;
                if (index) index[k] = x ^ 1ull;
                k++;
                i += 1;
            }
            nxt.push_back(x >> 1);
        }
        cur.swap(nxt);
    }
    *count = k;
    return DAPOL_OK;
}
// dapol_tree_build in TAPE mode: the padding nodes' blindings (Paddable::padding -> Scalar::random, src/dapol/node.rs:86-88) are read
// from `tape` -- tape_draws draws of 64 bytes, each reduced mod l, one per padding node in dapol_tree_padding_positions' order --
// instead of being derived from a seed.  With the draws a seed would give, the tree equals the seed-mode tree bit for bit.  A tree
// built from a tape has no seed to draw further padding nodes from: dapol_tree_update refuses it (build it again with a longer tape).
int32_t dapol_tree_build_tape(dapol_ctx* ctx, int32_t height, size_t n, const uint64_t* leaf_idx, const uint64_t* v, const uint8_t* r32,
                              const uint8_t* tape, size_t tape_draws, dapol_tree** out) {
    if (!ctx || !out || (n && (!leaf_idx || !v || !r32)) || (tape_draws && !tape)) return fail(DAPOL_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (height < 0 || height > 64) return fail(DAPOL_ERR_TREE_HEIGHT_TOO_BIG, "tree height must not exceed 64");
    if (n == 0) return fail(DAPOL_ERR_INVALID_ARGUMENT, "empty leaf set");
    if (n > ((size_t)1 << 31)) return fail(DAPOL_ERR_INVALID_ARGUMENT, "at most 2^31 leaves per GPU (32-bit node positions); memory is the practical bound");
    HIPCHK(hipSetDevice(ctx->device));
    dapol_tree_owned* t = new dapol_tree_owned();
    struct Guard { dapol_tree* t; ~Guard() { if (t) dapol_tree_destroy(t); } } guard{t};
    DevBuf<uint32_t> dtape;
    HIPCHK(dtape.alloc(tape_draws * 16 + 16));
    HIPCHK(t->leaves.idx.alloc(n)); HIPCHK(t->leaves.v.alloc(n)); HIPCHK(t->leaves.r.alloc(n * 8));
    HIPCHK(hipMemcpyAsync(t->leaves.idx.p, leaf_idx, n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(t->leaves.v.p, v, n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(t->leaves.r.p, r32, n * 32, hipMemcpyHostToDevice, ctx->stream));
    if (tape_draws) HIPCHK(hipMemcpyAsync(dtape.p, tape, tape_draws * 64, hipMemcpyHostToDevice, ctx->stream));
    const uint8_t no_seed[32] = {0};
    int32_t rc = tree_build_device(ctx, height, 0, n, t->leaves.idx.p, t->leaves.v.p, t->leaves.r.p, no_seed, t, dtape.p, tape_draws);
    if (rc != DAPOL_OK) return rc;
    guard.t = nullptr;
    t->holds_ctx = true;
    ctx_retain(ctx);
    *out = t;
    return DAPOL_OK;
}

int32_t dapol_tree_build_shard(dapol_ctx* ctx, int32_t total_height, int32_t shard_bits, size_t n, const uint64_t* leaf_idx,
                               const uint64_t* v, const uint8_t* r32, const uint8_t pad_seed32[32], dapol_tree** out) {
    if (!ctx || !out || !pad_seed32 || !n || !leaf_idx || !v || !r32) return fail(DAPOL_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (shard_bits) NEEDS_32_BYTE_DIGEST(ctx, "the sharded (multi-GPU) path");
    if (total_height < 0 || total_height > 64) return fail(DAPOL_ERR_TREE_HEIGHT_TOO_BIG, "tree height must not exceed 64");
    if (shard_bits < 0 || shard_bits > total_height || shard_bits > 16) return fail(DAPOL_ERR_INVALID_ARGUMENT, "shard_bits out of range");
    if (n > ((size_t)1 << 31)) return fail(DAPOL_ERR_INVALID_ARGUMENT, "at most 2^31 leaves per GPU (32-bit node positions); memory is the practical bound");
    HIPCHK(hipSetDevice(ctx->device));
    dapol_tree_owned* t = new dapol_tree_owned();
    struct Guard { dapol_tree* t; ~Guard() { if (t) dapol_tree_destroy(t); } } guard{t};
    HIPCHK(t->leaves.idx.alloc(n)); HIPCHK(t->leaves.v.alloc(n)); HIPCHK(t->leaves.r.alloc(n * 8));
    HIPCHK(hipMemcpyAsync(t->leaves.idx.p, leaf_idx, n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(t->leaves.v.p, v, n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(t->leaves.r.p, r32, n * 32, hipMemcpyHostToDevice, ctx->stream));
    int32_t rc = tree_build_device(ctx, total_height, shard_bits, n, t->leaves.idx.p, t->leaves.v.p, t->leaves.r.p, pad_seed32, t);
    if (rc != DAPOL_OK) return rc;
    guard.t = nullptr;
    t->holds_ctx = true;
    ctx_retain(ctx);
    *out = t;
    return DAPOL_OK;
}

// Dapol::update (src/dapol/mod.rs:211-213) for k leaves, applied in input order: a leaf is inserted, or replaces the
// one already at its index.  Padding nodes are keyed by position, so the updated tree is exactly what
// dapol_tree_build gives for the resulting leaf set; the host merges the (small) update into the sorted leaf arrays
// and the level-parallel build runs again -- one pass for the whole batch instead of k root-to-leaf walks.
// The incremental path of dapol_tree_update (kernels_ctx_tree.h, "incremental update"): every updated leaf already exists, so the
// tree keeps its structure and only the k root-to-leaf paths are re-merged, on the device, in three launches.  *done = false
// (nothing above the leaves touched) when some index is new: the caller then rebuilds.
static int32_t tree_update_incremental(dapol_tree_owned* own, size_t k, const std::vector<uint64_t>& idx, const std::vector<uint64_t>& v,
                                       const std::vector<uint8_t>& r, bool* done, std::vector<uint8_t>* found_out = nullptr) {
    *done = false;
    dapol_ctx* ctx = own->ctx;
    hipStream_t st = ctx->stream;
    const int H = own->height;
    // one staging buffer up, one scratch allocation (kept with the tree): idx | v | r | dv | dr | dP | pos | missing
    const size_t o_idx = 0, o_v = o_idx + k * 8, o_r = o_v + k * 8, o_dv = o_r + k * 32, o_dr = o_dv + k * 8, o_dP = o_dr + k * 32,
                 o_pos = o_dP + k * 160, o_miss = align_up(o_pos + k * (size_t)(H + 1) * 4, 8), o_found = o_miss + 8, total = o_found + k;
    if (own->upd_scratch.n < total) HIPCHK(own->upd_scratch.alloc(total + total / 2));
    std::vector<uint8_t> stage(o_dv);
    memcpy(stage.data() + o_idx, idx.data(), k * 8);
    memcpy(stage.data() + o_v, v.data(), k * 8);
    memcpy(stage.data() + o_r, r.data(), k * 32);
    uint8_t* d = own->upd_scratch.p;
    HIPCHK(hipMemcpyAsync(d, stage.data(), o_dv, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(d + o_miss, 0, 8, st));
    TreeUpdArgs U{k, H, (const uint64_t*)(d + o_idx), (const uint64_t*)(d + o_v), (const uint32_t*)(d + o_r), (uint32_t*)(d + o_pos), (int32_t*)(d + o_dP),
                  (uint64_t*)(d + o_dv), (uint32_t*)(d + o_dr), (uint32_t*)(d + o_miss), d + o_found, nullptr};
    hipLaunchKernelGGL(k_tree_upd_find, dim3(nblk(k, 64)), dim3(64), 0, st, own->d_views.p, U);
    LAUNCH_CHECK();
    uint32_t missing = 0;
    HIPCHK(hipMemcpyAsync(&missing, d + o_miss, 4, hipMemcpyDeviceToHost, st));
    if (found_out) { found_out->resize(k); HIPCHK(hipMemcpyAsync(found_out->data(), d + o_found, k, hipMemcpyDeviceToHost, st)); }
    HIPCHK(hipStreamSynchronize(st));
    if (missing) return DAPOL_OK;                            // a new index: nothing has been written; the caller inserts or rebuilds
    TreePoison poison{own, true};                            // from here on the leaves and the levels above are rewritten in place
    if (test_knob("DAPOL_TEST_FAIL_UPDATE_MIDWAY")) return fail(DAPOL_ERR_HIP, "injected failure between the leaf update and the re-merge (test knob)");
    hipLaunchKernelGGL(k_tree_upd_leaves, dim3(nblk(k, 64)), dim3(64), 0, st, ctx->tv, own->d_views.p, U);
    LAUNCH_CHECK();
    if (H >= 1) {
        hipLaunchKernelGGL(k_tree_upd_nodes, dim3(nblk(k * (size_t)H, 64)), dim3(64), 0, st, own->d_views.p, U);
        LAUNCH_CHECK();
        if (k <= 1024) {
            hipLaunchKernelGGL(k_tree_upd_hash, dim3(1), dim3((unsigned)align_up(k, 64)), 0, st, ctx->tv.digest, own->d_views.p, U, 0, H);
            LAUNCH_CHECK();
        } else {
            for (int lv = 0; lv < H; lv++) {
                hipLaunchKernelGGL(k_tree_upd_hash, dim3(nblk(k, 256)), dim3(256), 0, st, ctx->tv.digest, own->d_views.p, U, lv, lv + 1);
                LAUNCH_CHECK();
            }
        }
    }
    HIPCHK(hipStreamSynchronize(st));
    poison.armed = false;
    *done = true;
    return DAPOL_OK;
}

// Spans of one level inside a buffer of capacity `cap` records (incremental insert).
static size_t level_alt_bytes(size_t cap) { return align_up(cap * 8, 256) * 2 + align_up(cap * 32, 256) * 6 + align_up(cap * 4, 256) + align_up(cap, 256); }
static void level_alt_spans(uint8_t* base, size_t cap, LevelBuf& L, uint64_t** leaf_idx, uint64_t** leaf_v, uint32_t** leaf_r, bool is_leaf_level) {
    size_t o = 0;
    auto take = [&](size_t bytes) { uint8_t* p = base + o; o += align_up(bytes, 256); return p; };
    uint64_t* idx = (uint64_t*)take(cap * 8);
    uint64_t* vv = (uint64_t*)take(cap * 8);
    uint32_t* rr = (uint32_t*)take(cap * 32);
    if (is_leaf_level) { *leaf_idx = idx; *leaf_v = vv; *leaf_r = rr; }
    else { L.idx.p = idx; L.v.p = vv; L.r.p = rr; }
    L.C.p = (uint32_t*)take(cap * 32); L.H.p = (uint32_t*)take(cap * 32);
    L.padC.p = (uint32_t*)take(cap * 32); L.padH.p = (uint32_t*)take(cap * 32); L.padr.p = (uint32_t*)take(cap * 32);
    L.parent.p = (uint32_t*)take(cap * 4);
    L.has_pad.p = take(cap);
}

// The incremental path for NEW leaves (kernels_ctx_tree.h, "incremental insert"): k sorted, distinct indexes none of which is in the
// tree.  *done = false and nothing written when two new chains share a node (the caller rebuilds).
static int32_t tree_insert_incremental(dapol_tree_owned* own, size_t k, const std::vector<uint64_t>& idx, const std::vector<uint64_t>& v,
                                       const std::vector<uint8_t>& r, bool* done) {
    *done = false;
    dapol_ctx* ctx = own->ctx;
    hipStream_t st = ctx->stream;
    const int H = own->height;
    if (H < 1 || own->levels[0].n + k > ((size_t)1 << 31)) return DAPOL_OK;
    const size_t S1 = (size_t)H + 1;
    // scratch: idx | v | r | m | inspos | newpos | pos | dP | dv | dr | first | level insert positions | pad seed | flags
    const size_t o_idx = 0, o_v = o_idx + k * 8, o_r = o_v + k * 8, o_m = o_r + k * 32, o_ins = o_m + k * 4, o_new = o_ins + k * S1 * 4,
                 o_pos = o_new + k * S1 * 4, o_dP = align_up(o_pos + k * S1 * 4, 16), o_dv = o_dP + k * 160, o_dr = o_dv + k * 8, o_lvl = o_dr + k * 32,
                 o_seed = o_lvl + k * S1 * 4, o_flag = o_seed + 32, total = o_flag + 8;
    if (own->upd_scratch.n < total) HIPCHK(own->upd_scratch.alloc(total + total / 2));
    uint8_t* d = own->upd_scratch.p;
    {
        std::vector<uint8_t> stage(o_m);
        memcpy(stage.data() + o_idx, idx.data(), k * 8);
        memcpy(stage.data() + o_v, v.data(), k * 8);
        memcpy(stage.data() + o_r, r.data(), k * 32);
        HIPCHK(hipMemcpyAsync(d, stage.data(), o_m, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(d + o_seed, own->pad_seed, 32, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemsetAsync(d + o_flag, 0, 8, st));
    }
    TreeInsPlan P{k, H, (const uint64_t*)(d + o_idx), (uint32_t*)(d + o_m), (uint32_t*)(d + o_ins), (uint32_t*)(d + o_flag)};
    hipLaunchKernelGGL(k_tree_ins_plan, dim3(nblk(k, 64)), dim3(64), 0, st, own->d_views.p, P);
    LAUNCH_CHECK();
    std::vector<uint32_t> hm(k), hins(k * S1);
    uint32_t conflict = 0;
    HIPCHK(hipMemcpyAsync(hm.data(), d + o_m, k * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(hins.data(), d + o_ins, k * S1 * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&conflict, d + o_flag, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (conflict) return DAPOL_OK;                          // chains that share a node (or an index that exists): the rebuild handles it
    int max_m = 0;
    for (size_t j = 0; j < k; j++) max_m = std::max(max_m, (int)hm[j]);
    // per level t < max_m: the insert positions (old layout) of the chains that reach it, in leaf order
    std::vector<std::vector<uint32_t>> lvl((size_t)max_m + 1);
    std::vector<uint32_t> newpos(k * S1, 0), lvl_flat, lvl_off((size_t)max_m + 2, 0);
    for (size_t j = 0; j < k; j++)
        for (int t = 0; t < (int)hm[j]; t++) {
            newpos[j * S1 + t] = hins[j * S1 + t] + (uint32_t)lvl[t].size();
            lvl[t].push_back(hins[j * S1 + t]);
        }
    for (size_t j = 0; j < k; j++) {                         // the existing ancestor at level m_j, moved by what its level gains
        const int m = (int)hm[j];
        const uint32_t p = hins[j * S1 + m];
        uint32_t sh = 0;
        if (m < max_m) sh = (uint32_t)(std::upper_bound(lvl[m].begin(), lvl[m].end(), p) - lvl[m].begin());
        newpos[j * S1 + m] = p + sh;
    }
    for (int t = 0; t <= max_m; t++) { lvl_off[t] = (uint32_t)lvl_flat.size(); lvl_flat.insert(lvl_flat.end(), lvl[t].begin(), lvl[t].end()); }
    lvl_off[max_m + 1] = (uint32_t)lvl_flat.size();
    HIPCHK(hipMemcpyAsync(d + o_new, newpos.data(), k * S1 * 4, hipMemcpyHostToDevice, st));
    if (!lvl_flat.empty()) HIPCHK(hipMemcpyAsync(d + o_lvl, lvl_flat.data(), lvl_flat.size() * 4, hipMemcpyHostToDevice, st));
    // new storage for the levels that gain nodes, then move their existing nodes
    if (own->alt.size() != own->levels.size()) own->alt.resize(own->levels.size());
    std::vector<LevelBuf> newL(own->levels.begin(), own->levels.end());
    uint64_t *n_leaf_idx = own->leaf_idx, *n_leaf_v = own->leaf_v;
    uint32_t* n_leaf_r = own->leaf_r;
    std::vector<int> new_cur((size_t)max_m, -1);
    for (int t = 0; t < max_m; t++) {
        auto& A = own->alt[t];
        const int dstb = A.cur == 0 ? 1 : 0;
        const size_t n_new = own->levels[t].n + lvl[t].size();
        if (A.cap[dstb] < n_new) {
            const size_t cap = n_new + 4096 + n_new / 64;
            HIPCHK(A.buf[dstb].alloc(level_alt_bytes(cap)));
            A.cap[dstb] = cap;
        }
        level_alt_spans(A.buf[dstb].p, A.cap[dstb], newL[t], &n_leaf_idx, &n_leaf_v, &n_leaf_r, t == 0);
        newL[t].n = n_new;
        new_cur[t] = dstb;
    }
    auto view_of = [&](const std::vector<LevelBuf>& Ls, int t, uint64_t* li, uint64_t* lv_, uint32_t* lr) {
        const LevelBuf& L = Ls[t];
        LevelView lv;
        lv.n = L.n;
        lv.idx = t == 0 ? li : L.idx.p; lv.v = t == 0 ? lv_ : L.v.p; lv.r = t == 0 ? lr : L.r.p;
        lv.C = L.C.p; lv.H = L.H.p; lv.padC = L.padC.p; lv.padH = L.padH.p; lv.padr = L.padr.p; lv.has_pad = L.has_pad.p; lv.parent = L.parent.p; lv.ext = nullptr;
        return lv;
    };
    const uint32_t* d_lvl = (const uint32_t*)(d + o_lvl);
    for (int t = 0; t < max_m; t++) {
        LevelView src = view_of(std::vector<LevelBuf>(own->levels.begin(), own->levels.end()), t, own->leaf_idx, own->leaf_v, own->leaf_r);
        LevelView dst = view_of(newL, t, n_leaf_idx, n_leaf_v, n_leaf_r);
        const size_t n_old = own->levels[t].n;
        if (n_old)
            hipLaunchKernelGGL(k_tree_relayout, dim3(nblk(n_old, 256)), dim3(256), 0, st, src, dst, n_old, d_lvl + lvl_off[t], (uint32_t)lvl[t].size(),
                               d_lvl + lvl_off[t + 1], (uint32_t)lvl[t + 1].size(), 0);
        LAUNCH_CHECK();
    }
    // adopt the new storage, refresh the device-side views
    TreePoison poison{own, true};                            // the tree's own state changes from here on
    for (int t = 0; t < max_m; t++) { own->levels[t] = newL[t]; own->alt[t].cur = new_cur[t]; }
    own->leaf_idx = n_leaf_idx; own->leaf_v = n_leaf_v; own->leaf_r = n_leaf_r;
    std::vector<LevelView> hv((size_t)H + 1);
    for (int t = 0; t <= H; t++) hv[t] = own->view(t, nullptr);
    HIPCHK(hipMemcpyAsync(own->d_views.p, hv.data(), hv.size() * sizeof(LevelView), hipMemcpyHostToDevice, st));
    TreeInsArgs I{k, H, (const uint64_t*)(d + o_idx), (const uint64_t*)(d + o_v), (const uint32_t*)(d + o_r), (const uint32_t*)(d + o_m),
                  (const uint32_t*)(d + o_new), (uint32_t*)(d + o_pos), (int32_t*)(d + o_dP), (uint64_t*)(d + o_dv), (uint32_t*)(d + o_dr),
                  (const uint32_t*)(d + o_seed)};
    hipLaunchKernelGGL(k_tree_ins_chain, dim3((unsigned)k), dim3(64), 0, st, ctx->tv, own->d_views.p, I);
    LAUNCH_CHECK();
    TreeUpdArgs U{k, H, (const uint64_t*)(d + o_idx), (const uint64_t*)(d + o_v), (const uint32_t*)(d + o_r), (uint32_t*)(d + o_pos), (int32_t*)(d + o_dP),
                  (uint64_t*)(d + o_dv), (uint32_t*)(d + o_dr), (uint32_t*)(d + o_flag), nullptr, (const uint32_t*)(d + o_m)};
    hipLaunchKernelGGL(k_tree_upd_nodes, dim3(nblk(k * (size_t)H, 64)), dim3(64), 0, st, own->d_views.p, U);
    LAUNCH_CHECK();
    if (k <= 1024) {
        hipLaunchKernelGGL(k_tree_upd_hash, dim3(1), dim3((unsigned)align_up(k, 64)), 0, st, ctx->tv.digest, own->d_views.p, U, 0, H);
        LAUNCH_CHECK();
    } else {
        for (int lv = 0; lv < H; lv++) {
            hipLaunchKernelGGL(k_tree_upd_hash, dim3(nblk(k, 256)), dim3(256), 0, st, ctx->tv.digest, own->d_views.p, U, lv, lv + 1);
            LAUNCH_CHECK();
        }
    }
    HIPCHK(hipStreamSynchronize(st));
    for (size_t j = 0; j < k; j++) { own->n_real += hm[j]; own->n_pad += (uint64_t)hm[j] - 2 + 0; }     // m - 1 new padding nodes, one dropped
    poison.armed = false;
    *done = true;
    return DAPOL_OK;
}

static int32_t tree_update_impl(dapol_tree* tree, size_t k, const uint64_t* leaf_idx, const uint64_t* v, const uint8_t* r32);
int32_t dapol_tree_update(dapol_tree* tree, size_t k, const uint64_t* leaf_idx, const uint64_t* v, const uint8_t* r32) {
    int32_t rc = tree_update_impl(tree, k, leaf_idx, v, r32);
    if (rc || !tree || !k || !ctx_wide(tree->ctx)) return rc;
    // a 64-byte digest: the in-place paths re-hash 32-byte chains only; lay the whole 64-byte chain again (the rebuild path has
    // done so already -- once more costs a pass over the nodes and keeps this wrapper free of cases)
    rc = tree_hash_wide(tree);
    if (rc) tree->invalid = true;
    return rc;
}
static int32_t tree_update_impl(dapol_tree* tree, size_t k, const uint64_t* leaf_idx, const uint64_t* v, const uint8_t* r32) {
    if (!tree || (k && (!leaf_idx || !v || !r32))) return fail(DAPOL_ERR_INVALID_ARGUMENT, "null argument");
    if (k == 0) return DAPOL_OK;
    TREE_USABLE(tree);
    if (tree->tape_built) return fail(DAPOL_ERR_INVALID_ARGUMENT, "the tree was built from a padding tape: an update may need draws the tape does not hold; build it again");
    dapol_tree_owned* own = static_cast<dapol_tree_owned*>(tree);
    if (!own->leaves.idx.p) return fail(DAPOL_ERR_INVALID_ARGUMENT, "tree does not own its leaves (workload tree): rebuild the workload instead");
    dapol_ctx* ctx = tree->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    // Replacing the liabilities of leaves that exist (what smtree's update does to ONE leaf: re-merge its root-to-leaf path) keeps the
    // structure: up to DAPOL_UPDATE_INCREMENTAL_MAX (default 65,536, and at most an eighth of the leaves) such updates are applied
    // in place on the device.  Anything else -- a new index, a big batch -- takes the rebuild below, which is bit for bit the same tree.
    {
        size_t inc_max = 65536;
        if (ctx->opt.update_incremental_max > 0) inc_max = (size_t)ctx->opt.update_incremental_max;
        if (ctx->opt.update_incremental_max < 0) inc_max = 0;
        if (const char* e = knob("DAPOL_UPDATE_INCREMENTAL_MAX")) inc_max = (size_t)atoll(e);
        if (k <= inc_max && k <= tree->levels[0].n / 8 + 1 && tree->levels[0].n > 0) {
            std::vector<uint32_t> ord(k);
            for (size_t i = 0; i < k; i++) ord[i] = (uint32_t)i;
            std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return leaf_idx[a] < leaf_idx[b]; });
            std::vector<uint64_t> si, sv;
            std::vector<uint8_t> sr;
            for (size_t b = 0; b < k; b++) {
                if (b + 1 < k && leaf_idx[ord[b + 1]] == leaf_idx[ord[b]]) continue;      // later updates of the same index win
                const uint32_t u = ord[b];
                si.push_back(leaf_idx[u]); sv.push_back(v[u]);
                sr.insert(sr.end(), r32 + (size_t)u * 32, r32 + (size_t)u * 32 + 32);
            }
            bool done = false;
            std::vector<uint8_t> found;
            int32_t rc = tree_update_incremental(own, si.size(), si, sv, sr, &done, &found);
            if (rc != DAPOL_OK) return rc;
            if (done) { own->last_update_path = 1; return DAPOL_OK; }
            // Some indexes are new.  Up to 4,096 new leaves whose new chains are disjoint are inserted in place (the level arrays
            // that gain nodes are rewritten in order, nothing is recomputed for the nodes that merely move); then the leaves that
            // did exist are replaced as above.  Otherwise nothing has been written and the rebuild below takes the whole batch.
            std::vector<uint64_t> ni, nv, ei, ev;
            std::vector<uint8_t> nr, er;
            for (size_t b = 0; b < si.size(); b++) {
                auto& di = found[b] ? ei : ni; auto& dv_ = found[b] ? ev : nv; auto& dr_ = found[b] ? er : nr;
                di.push_back(si[b]); dv_.push_back(sv[b]);
                dr_.insert(dr_.end(), sr.begin() + b * 32, sr.begin() + b * 32 + 32);
            }
            const int H = tree->height;
            bool in_range = true;
            for (uint64_t x : ni) if (tree->index_bits < 64 && (x >> tree->index_bits)) in_range = false;         // (the rebuild reports the error)
            if (tree->shard_bits && in_range && !ni.empty()) {                                                   // shard trees: a new leaf must carry the shard's prefix (else the rebuild reports the error)
                uint64_t first = 0;
                HIPCHK(hipMemcpy(&first, tree->leaf_idx, 8, hipMemcpyDeviceToHost));
                const int sh = tree->index_bits - tree->shard_bits;
                for (uint64_t x : ni) if ((x >> sh) != (first >> sh)) in_range = false;
            }
            if (!ni.empty() && ni.size() <= 4096 && in_range && H >= 1 && !knob("DAPOL_NO_INCREMENTAL_INSERT")) {
                rc = tree_insert_incremental(own, ni.size(), ni, nv, nr, &done);
                if (rc != DAPOL_OK) return rc;
                if (done) {
                    own->last_update_path = 2;
                    if (ei.empty()) return DAPOL_OK;
                    rc = tree_update_incremental(own, ei.size(), ei, ev, er, &done);
                    //  the inserts are in: whatever stops the replacements now -- also an error BEFORE their first write,
                    // which would leave a consistent but half-updated tree -- is a failed in-place update; the tree is refused from here on
                    if (rc != DAPOL_OK) { own->invalid = true; return rc; }
                    if (done) { own->last_update_path = 3; return DAPOL_OK; }
                    own->invalid = true;                     // the new leaves are in, the replacements are not
                    return fail(DAPOL_ERR_INVALID_ARGUMENT, "internal: a leaf found before the insert was not found after it");
                }
            }
        }
    }
    const size_t n0 = tree->levels[0].n;
    std::vector<uint64_t> oi(n0), ov(n0);
    std::vector<uint8_t> orr(n0 * 32);
    HIPCHK(hipMemcpyAsync(oi.data(), tree->leaf_idx, n0 * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(ov.data(), tree->leaf_v, n0 * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(orr.data(), tree->leaf_r, n0 * 32, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    std::vector<uint32_t> ord(k);
    for (size_t i = 0; i < k; i++) ord[i] = (uint32_t)i;
    std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return leaf_idx[a] < leaf_idx[b]; });
    std::vector<uint64_t> ni, nv;
    std::vector<uint8_t> nr;
    ni.reserve(n0 + k); nv.reserve(n0 + k); nr.reserve((n0 + k) * 32);
    size_t a = 0, b = 0;
    while (a < n0 || b < k) {
        if (b < k) {                                        // later updates of the same index win
            while (b + 1 < k && leaf_idx[ord[b + 1]] == leaf_idx[ord[b]]) b++;
        }
        bool take_new = b < k && (a >= n0 || leaf_idx[ord[b]] <= oi[a]);
        if (take_new) {
            uint32_t u = ord[b];
            if (a < n0 && oi[a] == leaf_idx[u]) a++;         // replaces
            ni.push_back(leaf_idx[u]); nv.push_back(v[u]);
            nr.insert(nr.end(), r32 + (size_t)u * 32, r32 + (size_t)u * 32 + 32);
            b++;
        } else {
            ni.push_back(oi[a]); nv.push_back(ov[a]);
            nr.insert(nr.end(), orr.begin() + a * 32, orr.begin() + a * 32 + 32);
            a++;
        }
    }
    const size_t n = ni.size();
    if (n > ((size_t)1 << 31)) return fail(DAPOL_ERR_INVALID_ARGUMENT, "at most 2^31 leaves per GPU (32-bit node positions); memory is the practical bound");
    dapol_tree_owned fresh;
    HIPCHK(fresh.leaves.idx.alloc(n)); HIPCHK(fresh.leaves.v.alloc(n)); HIPCHK(fresh.leaves.r.alloc(n * 8));
    HIPCHK(hipMemcpyAsync(fresh.leaves.idx.p, ni.data(), n * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(fresh.leaves.v.p, nv.data(), n * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(fresh.leaves.r.p, nr.data(), n * 32, hipMemcpyHostToDevice, st));
    uint8_t seed[32];
    memcpy(seed, tree->pad_seed, 32);
    int32_t rc = tree_build_device(ctx, tree->index_bits, tree->shard_bits, n, fresh.leaves.idx.p, fresh.leaves.v.p, fresh.leaves.r.p, seed, &fresh);
    if (rc != DAPOL_OK) return rc;                          // the old tree stays as it was
    fresh.holds_ctx = own->holds_ctx;
    *own = std::move(fresh);
    own->last_update_path = 0;
    return DAPOL_OK;
}
// What the last dapol_tree_update on this tree did: 0 = rebuilt the tree, 1 = replaced existing leaves in place, 2 = inserted new
// leaves in place, 3 = both.  (Diagnostics: the result is the same tree whichever path ran.)
int32_t dapol_tree_last_update_path(dapol_tree* tree, int32_t* path) {
    if (!tree || !path) return fail(DAPOL_ERR_INVALID_ARGUMENT, "null argument");
    *path = static_cast<dapol_tree_owned*>(tree)->last_update_path;
    return DAPOL_OK;
}

// Paddable::padding (src/dapol/node.rs:86-88): new(0, random blinding) with the draw keyed by the node's position.
__global__ void k_padding_nodes(TableView tbl, size_t n, const uint32_t* pad_seed, const uint8_t* level, const uint64_t* index, uint32_t* C,
                                uint32_t* H, uint32_t* r) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t seed[8], wide[16], rB[8], cB[8], hB[8];
    for (int k = 0; k < 8; k++) seed[k] = pad_seed[k];
    seed_wide(wide, seed, 1u, (uint64_t)level[i], index[i]);
    sc rm;
    sc_from_wide(rm, wide);
    sc_from_mont(rB, rm);
    ge_p3 p;
    ge_identity(p);
    tbl_fixed_mul_add(p, tbl, tbl.row_Bb(0), rB);
    ge_compress(cB, p);
    node_hash32(tbl.digest, hB, cB);
    st8(C + i * 8, cB);
    st8(H + i * 8, hB);
    st8(r + i * 8, rB);
}
int32_t dapol_padding_nodes(dapol_ctx* ctx, const uint8_t pad_seed32[32], size_t n, const uint8_t* level, const uint64_t* index, uint8_t* C32,
                            uint8_t* H32, uint8_t* r32) {
    if (!ctx || !pad_seed32 || (n && (!level || !index || !C32 || !H32 || !r32))) return fail(DAPOL_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return DAPOL_OK;
    for (size_t i = 0; i < n; i++)
        if (level[i] > 64) return fail(DAPOL_ERR_INVALID_ARGUMENT, "level must not exceed 64");
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    DevBuf<uint32_t> seed, dC, dH, dr;
    DevBuf<uint8_t> dl;
    DevBuf<uint64_t> di;
    HIPCHK(seed.alloc(8)); HIPCHK(dC.alloc(n * 8)); HIPCHK(dH.alloc(n * 8)); HIPCHK(dr.alloc(n * 8)); HIPCHK(dl.alloc(n)); HIPCHK(di.alloc(n));
    HIPCHK(hipMemcpyAsync(seed.p, pad_seed32, 32, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dl.p, level, n, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(di.p, index, n * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_padding_nodes, dim3(nblk(n, 64)), dim3(64), 0, st, ctx->tv, n, seed.p, dl.p, di.p, dC.p, dH.p, dr.p);
    LAUNCH_CHECK();
    DevBuf<uint32_t> dHw;
    if (ctx_wide(ctx)) {
        HIPCHK(dHw.alloc(n * 16));
        hipLaunchKernelGGL(k_wide_hash_leaves, dim3(nblk(n, 256)), dim3(256), 0, st, n, dC.p, dHw.p);
        LAUNCH_CHECK();
    }
    HIPCHK(hipMemcpyAsync(C32, dC.p, n * 32, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(H32, ctx_wide(ctx) ? dHw.p : dH.p, n * ctx_hash_bytes(ctx), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(r32, dr.p, n * 32, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return DAPOL_OK;
}

// Mergeable::merge on compressed records
template <int HW>
__device__ __forceinline__ void ldh(uint32_t* w, const uint32_t* p) { ld8(w, p); if (HW == 16) ld8(w + 8, p + 8); }
template <int HW>
__device__ __forceinline__ void sth(uint32_t* p, const uint32_t* w) { st8(p, w); if (HW == 16) st8(p + 8, w + 8); }
template <int HW>
__global__ void k_merge_records(int dg, size_t n, const uint32_t* CL, const uint32_t* HL, const uint64_t* vL, const uint32_t* rL,
                                const uint32_t* CR, const uint32_t* HR, const uint64_t* vR, const uint32_t* rR, uint32_t* C, uint32_t* H,
                                uint64_t* v, uint32_t* r, uint32_t* bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t cl[8], cr[8], hl[HW], hr[HW], cp[8], hp[HW];
    ld8(cl, CL + i * 8); ld8(cr, CR + i * 8); ldh<HW>(hl, HL + i * HW); ldh<HW>(hr, HR + i * HW);
    ge_p3 a, b, p;
    bool ok = ge_decompress(a, cl) & ge_decompress(b, cr);
    if (!ok) { atomicOr(bad, 1u); return; }
    ge_add(p, a, b);
    ge_compress(cp, p);
    node_hash_parent_w<HW>(dg, hp, cl, cr, hl, hr);
    st8(C + i * 8, cp);
    sth<HW>(H + i * HW, hp);
    if (v) {
        uint32_t ra[8], rb[8], rp[8];
        ld8(ra, rL + i * 8); ld8(rb, rR + i * 8);
        sc ma, mb, ms;
        sc_to_mont(ma, ra); sc_to_mont(mb, rb); sc_add(ms, ma, mb); sc_from_mont(rp, ms);
        st8(r + i * 8, rp);
        v[i] = vL[i] + vR[i];
    }
}

int32_t dapol_merge_batch(dapol_ctx* ctx, size_t n, const uint8_t* CL32, const uint8_t* HL32, const uint64_t* vL, const uint8_t* rL32,
                          const uint8_t* CR32, const uint8_t* HR32, const uint64_t* vR, const uint8_t* rR32, uint8_t* C32, uint8_t* H32,
                          uint64_t* v, uint8_t* r32) {
    if (!ctx || (n && (!CL32 || !HL32 || !CR32 || !HR32 || !C32 || !H32))) return fail(DAPOL_ERR_INVALID_ARGUMENT, "null argument");
    bool with_secrets = vL || rL32 || vR || rR32 || v || r32;
    if (with_secrets && !(vL && rL32 && vR && rR32 && v && r32)) return fail(DAPOL_ERR_INVALID_ARGUMENT, "v/r pointers must be all set or all null");
    if (n == 0) return DAPOL_OK;
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    DevBuf<uint32_t> d[8], bad;
    DevBuf<uint64_t> dv[3];
    const size_t hw = (size_t)ctx_hw(ctx), hb = hw * 4;            // H arrays: ctx_hash_bytes per node
    for (auto& x : d) HIPCHK(x.alloc(n * hw));
    HIPCHK(bad.alloc(1));
    HIPCHK(hipMemsetAsync(bad.p, 0, 4, st));
    HIPCHK(hipMemcpyAsync(d[0].p, CL32, n * 32, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d[1].p, HL32, n * hb, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d[2].p, CR32, n * 32, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d[3].p, HR32, n * hb, hipMemcpyHostToDevice, st));
    if (with_secrets) {
        for (auto& x : dv) HIPCHK(x.alloc(n));
        HIPCHK(hipMemcpyAsync(d[4].p, rL32, n * 32, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(d[5].p, rR32, n * 32, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(dv[0].p, vL, n * 8, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(dv[1].p, vR, n * 8, hipMemcpyHostToDevice, st));
    }
    DevBuf<uint32_t> oC, oH;
    HIPCHK(oC.alloc(n * 8)); HIPCHK(oH.alloc(n * hw));
    if (hw == 16)
        hipLaunchKernelGGL(k_merge_records<16>, dim3(nblk(n, 64)), dim3(64), 0, st, ctx->tv.digest, n, d[0].p, d[1].p, with_secrets ? dv[0].p : nullptr, d[4].p,
                           d[2].p, d[3].p, with_secrets ? dv[1].p : nullptr, d[5].p, oC.p, oH.p, with_secrets ? dv[2].p : nullptr, d[6].p, bad.p);
    else
        hipLaunchKernelGGL(k_merge_records<8>, dim3(nblk(n, 64)), dim3(64), 0, st, ctx->tv.digest, n, d[0].p, d[1].p, with_secrets ? dv[0].p : nullptr, d[4].p,
                           d[2].p, d[3].p, with_secrets ? dv[1].p : nullptr, d[5].p, oC.p, oH.p, with_secrets ? dv[2].p : nullptr, d[6].p, bad.p);
    LAUNCH_CHECK();
    uint32_t h_bad = 0;
    HIPCHK(hipMemcpyAsync(&h_bad, bad.p, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(C32, oC.p, n * 32, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(H32, oH.p, n * hb, hipMemcpyDeviceToHost, st));
    if (with_secrets) {
        HIPCHK(hipMemcpyAsync(v, dv[2].p, n * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(r32, d[6].p, n * 32, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    if (h_bad) return fail(DAPOL_ERR_VALUE_DECODING, "Not the canonical encoding of a point.");
    return DAPOL_OK;
}

int32_t dapol_tree_destroy(dapol_tree* tree) {
    if (!tree) return DAPOL_OK;
    dapol_tree_owned* own = static_cast<dapol_tree_owned*>(tree);
    dapol_ctx* ctx = own->holds_ctx ? tree->ctx : nullptr;
    if (tree->ctx) (void)hipSetDevice(tree->ctx->device);
    delete own;
    if (ctx) (void)dapol_ctx_destroy(ctx);
    return DAPOL_OK;
}

int32_t dapol_tree_root(dapol_tree* tree, uint8_t C32[32], uint8_t H32[32], uint64_t* v, uint8_t r32[32]) {
    if (!tree) return fail(DAPOL_ERR_INVALID_ARGUMENT, "null tree");
    TREE_USABLE(tree);
    HIPCHK(hipSetDevice(tree->ctx->device));
    LevelView lv = tree->view(tree->height, nullptr);
    if (C32) HIPCHK(hipMemcpy(C32, lv.C, 32, hipMemcpyDeviceToHost));
    if (H32) HIPCHK(hipMemcpy(H32, ctx_wide(tree->ctx) ? tree->wviews[tree->height].H : lv.H, ctx_hash_bytes(tree->ctx), hipMemcpyDeviceToHost));
    if (v) HIPCHK(hipMemcpy(v, lv.v, 8, hipMemcpyDeviceToHost));
    if (r32) HIPCHK(hipMemcpy(r32, lv.r, 32, hipMemcpyDeviceToHost));
    return DAPOL_OK;
}

int32_t dapol_tree_node_count(dapol_tree* tree, uint64_t* real_nodes, uint64_t* padding_nodes) {
    if (!tree) return fail(DAPOL_ERR_INVALID_ARGUMENT, "null tree");
    TREE_USABLE(tree);
    if (real_nodes) *real_nodes = tree->n_real;
    if (padding_nodes) *padding_nodes = tree->n_pad;
    return DAPOL_OK;
}

static int32_t level_pad_flags(dapol_tree* tree, int level, std::vector<uint8_t>& hp) {
    LevelView lv = tree->view(level, nullptr);
    hp.resize(lv.n);
    if (level == tree->height) { std::fill(hp.begin(), hp.end(), 0); return DAPOL_OK; }
    HIPCHK(hipMemcpy(hp.data(), lv.has_pad, lv.n, hipMemcpyDeviceToHost));
    return DAPOL_OK;
}

int32_t dapol_tree_level_size(dapol_tree* tree, int32_t level, uint64_t* n_real, uint64_t* n_pad) {
    if (!tree || level < 0 || level > tree->height) return fail(DAPOL_ERR_INVALID_ARGUMENT, "bad level");
    TREE_USABLE(tree);
    HIPCHK(hipSetDevice(tree->ctx->device));
    std::vector<uint8_t> hp;
    int32_t rc = level_pad_flags(tree, level, hp);
    if (rc) return rc;
    uint64_t np = 0;
    for (uint8_t f : hp) np += f;
    if (n_real) *n_real = hp.size();
    if (n_pad) *n_pad = np;
    return DAPOL_OK;
}

int32_t dapol_tree_level_nodes(dapol_tree* tree, int32_t level, uint64_t* idx, uint64_t* v, uint8_t* r32, uint8_t* C32, uint8_t* H32,
                               uint8_t* is_pad) {
    if (!tree || level < 0 || level > tree->height || !idx || !v || !r32 || !C32 || !H32 || !is_pad)
        return fail(DAPOL_ERR_INVALID_ARGUMENT, "bad argument");
    TREE_USABLE(tree);
    HIPCHK(hipSetDevice(tree->ctx->device));
    LevelView lv = tree->view(level, nullptr);
    size_t n = lv.n;
    std::vector<uint8_t> hp;
    int32_t rc = level_pad_flags(tree, level, hp);
    if (rc) return rc;
    HIPCHK(hipMemcpy(idx, lv.idx, n * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(v, lv.v, n * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(r32, lv.r, n * 32, hipMemcpyDeviceToHost));
    const bool wide = ctx_wide(tree->ctx);
    const size_t hb = ctx_hash_bytes(tree->ctx);
    HIPCHK(hipMemcpy(C32, lv.C, n * 32, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(H32, wide ? tree->wviews[level].H : lv.H, n * hb, hipMemcpyDeviceToHost));
    memset(is_pad, 0, n);
    std::vector<uint8_t> pc(n * 32), ph(n * hb), pr(n * 32);
    if (level < tree->height) {
        HIPCHK(hipMemcpy(pc.data(), lv.padC, n * 32, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(ph.data(), wide ? tree->wviews[level].padH : lv.padH, n * hb, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(pr.data(), lv.padr, n * 32, hipMemcpyDeviceToHost));
    }
    size_t o = n;
    for (size_t i = 0; i < n; i++) {
        if (!hp[i]) continue;
        idx[o] = idx[i] ^ 1ull;
        v[o] = 0;
        memcpy(r32 + o * 32, pr.data() + i * 32, 32);
        memcpy(C32 + o * 32, pc.data() + i * 32, 32);
        memcpy(H32 + o * hb, ph.data() + i * hb, hb);
        is_pad[o] = 1;
        o++;
    }
    return DAPOL_OK;
}

// Gathers the siblings of b leaves into device buffers (any of which may be null).
static int32_t tree_paths_device(dapol_tree* tree, size_t b, const uint64_t* d_leaf_idx, PathOut out, uint32_t* d_pos, int n_upper = 0) {
    TREE_USABLE(tree);
    hipStream_t st = tree->ctx->stream;
    DevBuf<uint32_t> missing;
    HIPCHK(missing.alloc(1));
    HIPCHK(hipMemsetAsync(missing.p, 0, 4, st));
    LevelView l0 = tree->view(0, nullptr);
    hipLaunchKernelGGL(k_tree_find_leaves, dim3(nblk(b, 256)), dim3(256), 0, st, b, d_leaf_idx, l0.n, l0.idx, d_pos, missing.p);
    LAUNCH_CHECK();
    uint32_t h_missing = 0;
    HIPCHK(hipMemcpyAsync(&h_missing, missing.p, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (h_missing) return fail(DAPOL_ERR_UNKNOWN_LEAF, "no liability at one of the requested leaves");
    if (ctx_wide(tree->ctx) && out.H) {            // 64-byte hashes: out.H is [b][height + n_upper][16], filled from the wide chain
        if (tree->height) {
            hipLaunchKernelGGL(k_wide_path_walk, dim3(nblk(b, 64)), dim3(64), 0, st, b, d_pos, tree->d_views.p, tree->d_wviews.p, tree->height, n_upper,
                               g_wire.siblings_leaf_first, out.H);
            LAUNCH_CHECK();
        }
        out.H = nullptr;
    }
    if (tree->height) {
        hipLaunchKernelGGL(k_tree_path_walk, dim3(nblk(b, 64)), dim3(64), 0, st, b, d_pos, tree->d_views.p, tree->height, n_upper, g_wire.siblings_leaf_first, out);
        LAUNCH_CHECK();
    }
    return DAPOL_OK;
}

int32_t dapol_tree_paths(dapol_tree* tree, size_t b, const uint64_t* leaf_idx, uint8_t* sib_C32, uint8_t* sib_H32, uint64_t* sib_v,
                         uint8_t* sib_r32) {
    WIRE_SCOPE();
    if (!tree || (b && !leaf_idx)) return fail(DAPOL_ERR_INVALID_ARGUMENT, "null argument");
    if (b == 0) return DAPOL_OK;
    HIPCHK(hipSetDevice(tree->ctx->device));
    hipStream_t st = tree->ctx->stream;
    size_t h = (size_t)tree->height, tot = b * h;
    DevBuf<uint64_t> dl, dv;
    DevBuf<uint32_t> dC, dH, dr, dpos;
    HIPCHK(dl.alloc(b)); HIPCHK(dpos.alloc(b));
    HIPCHK(dC.alloc(tot * 8)); HIPCHK(dH.alloc(tot * (size_t)ctx_hw(tree->ctx))); HIPCHK(dr.alloc(tot * 8)); HIPCHK(dv.alloc(tot));
    HIPCHK(hipMemcpyAsync(dl.p, leaf_idx, b * 8, hipMemcpyHostToDevice, st));
    PathOut po{dC.p, dH.p, dv.p, dr.p};
    int32_t rc = tree_paths_device(tree, b, dl.p, po, dpos.p);
    if (rc) return rc;
    if (sib_C32) HIPCHK(hipMemcpyAsync(sib_C32, dC.p, tot * 32, hipMemcpyDeviceToHost, st));
    if (sib_H32) HIPCHK(hipMemcpyAsync(sib_H32, dH.p, tot * ctx_hash_bytes(tree->ctx), hipMemcpyDeviceToHost, st));
    if (sib_v) HIPCHK(hipMemcpyAsync(sib_v, dv.p, tot * 8, hipMemcpyDeviceToHost, st));
    if (sib_r32) HIPCHK(hipMemcpyAsync(sib_r32, dr.p, tot * 32, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return DAPOL_OK;
}

#include "host_range.inc"
#include "host_leaf.inc"
#include "host_wire.inc"
#include "host_batch.inc"
#include "host_comm.inc"
