// Leaf derivation on the device: build_leaf_nodes / shuffle_index (src/dapol/mod.rs:323-441).
//   audit_id   = D(audit_seed || internal_id)                                   (:347-353)
//   index_seed = D(audit_id || "index_seed" || external_id)                     (:361-368)
//   index      = first of  idx_k = be64(D^k(index_seed)[..8]) >> (64 - height), k = 1..128, not taken yet (:408-441)
//   blinding   = Scalar::from_bits(D(audit_id || "blind_seed" || external_id))  (:377-385)
// The hashing is embarrassingly parallel.  "Not taken yet" is order dependent in the reference (an entity only
// competes with the entities BEFORE it in the input).  That order is kept WITHOUT walking the entities one by one
// (k_leaf_claim_first / k_leaf_settle below): every entity claims its candidate in a table of slots whose owner can
// only become an EARLIER entity (atomicMin on the input position); an entity that finds its slot owned by an earlier
// one moves to its next candidate; repeat until nobody moves.  Every move is forced by an earlier entity that holds
// the slot for good (it leaves only for one earlier still), so each entity walks a prefix of the walk the sequential
// loop gives it, and the fixed point is the sequential result -- also for the failing entity of FailedToMapIndex
// (the earliest one that runs out of its 128 candidates).  Near the sparsity bound Dapol::new allows (2^height = 2 n)
// almost half of the entities collide: the one-lane resolution this replaces took 6.7 s for 2^20 liabilities at
// height 21 (profiles/r04l_leaf_bound.txt); it remains for height 64 (whose indexes leave no spare key for "empty").
#pragma once
#include <hip/hip_runtime.h>
#include "hash.h"

namespace dapol {

enum { LEAF_MAX_RETRIES = 128 };            // MAX_INDEX_RETRIES, src/dapol/mod.rs:29

struct LeafArgs {
    size_t n;
    int height, kind;
    const uint8_t* seed; uint32_t seed_len;
    const uint8_t* iid; const uint32_t* iid_off;     // [n+1] byte offsets
    const uint8_t* eid; const uint32_t* eid_off;
    uint32_t* audit_id;      // [n][8]
    uint32_t* idx_state;     // [n][8]  D^k(index_seed) of the candidate currently held
    uint64_t* cand;          // [n]     candidate / final index
    uint32_t* blind;         // [n][8]
    uint32_t* err;           // [0] = digest overflow flag, [1] = duplicate id flag, [2] = failed-to-map flag, [3] = offending entity
    int max_tries;           // MAX_INDEX_RETRIES = 128.  Lower only under the test knob DAPOL_LEAF_MAX_TRIES: with 2^height >= 2 n the
                             // reference's FailedToMapIndex has probability < 2^-128 per entity, so no input reaches it at 128 tries
};

__device__ __forceinline__ uint64_t leaf_index_of(const uint32_t* st, int height) {
    uint64_t be = ((uint64_t)__builtin_bswap32(st[0]) << 32) | (uint64_t)__builtin_bswap32(st[1]);
    return be >> (64 - height);
}

__global__ __launch_bounds__(64) void k_leaf_hash(LeafArgs A) {
    size_t e = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (e >= A.n) return;
    Digest d;
    uint32_t aid[8], st[8], bl[8];
    uint32_t stack[B3_STACK_DEPTH * 8];                  // ids of any length (mod.rs:347-349, 358-360): BLAKE3 beyond one chunk chains
                                                         // its chunks through this stack (private memory, touched by long inputs only)
    dg_init_long(d, A.kind, stack);
    dg_update(d, A.seed, A.seed_len);
    dg_update(d, A.iid + A.iid_off[e], A.iid_off[e + 1] - A.iid_off[e]);
    bool ovf = d.overflow;
    dg_final(d, aid);
    const uint8_t t1[10] = {'i', 'n', 'd', 'e', 'x', '_', 's', 'e', 'e', 'd'};
    const uint8_t t2[10] = {'b', 'l', 'i', 'n', 'd', '_', 's', 'e', 'e', 'd'};
    dg_init_long(d, A.kind, stack);
    dg_update_words(d, aid, 8);
    dg_update(d, t1, 10);
    dg_update(d, A.eid + A.eid_off[e], A.eid_off[e + 1] - A.eid_off[e]);
    ovf |= d.overflow;
    dg_final(d, st);
    dg_init(d, A.kind);                                  // first shuffle_index iteration: seed = D(seed)
    dg_update_words(d, st, 8);
    dg_final(d, st);
    dg_init_long(d, A.kind, stack);
    dg_update_words(d, aid, 8);
    dg_update(d, t2, 10);
    dg_update(d, A.eid + A.eid_off[e], A.eid_off[e + 1] - A.eid_off[e]);
    ovf |= d.overflow;
    dg_final(d, bl);
    bl[7] &= 0x7fffffffu;                                // Scalar::from_bits
    for (int i = 0; i < 8; i++) { A.audit_id[e * 8 + i] = aid[i]; A.idx_state[e * 8 + i] = st[i]; A.blind[e * 8 + i] = bl[i]; }
    A.cand[e] = leaf_index_of(st, A.height);
    if (ovf) atomicOr(&A.err[0], 1u);
}

// Duplicate internal ids (src/dapol/mod.rs:341-343): entities sorted by the first 8 bytes of audit_id; equal
// neighbours are compared in full (audit id, then the id bytes themselves).
__global__ void k_leaf_dup_keys(size_t n, const uint32_t* audit_id, uint64_t* key, uint32_t* ord) {
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    key[e] = ((uint64_t)audit_id[e * 8 + 1] << 32) | audit_id[e * 8];
    ord[e] = (uint32_t)e;
}
__global__ void k_leaf_dup_check(LeafArgs A, const uint64_t* key, const uint32_t* ord) {
    size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p == 0 || p >= A.n || key[p] != key[p - 1]) return;
    uint32_t a = ord[p - 1], b = ord[p];
    uint32_t la = A.iid_off[a + 1] - A.iid_off[a], lb = A.iid_off[b + 1] - A.iid_off[b];
    if (la != lb) return;
    for (uint32_t i = 0; i < la; i++)
        if (A.iid[A.iid_off[a] + i] != A.iid[A.iid_off[b] + i]) return;
    atomicOr(&A.err[1], 1u);
    atomicMax(&A.err[3], a > b ? a : b);
}

// Collision bookkeeping.  After sorting (cand, input order) pairs: the first entity of every equal-candidate group
// holds the slot; the others are "losers".
__global__ void k_leaf_mark(size_t n, const uint64_t* skey, const uint32_t* sord, uint32_t* losers, uint32_t* n_losers) {
    size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p == 0 || p >= n || skey[p] != skey[p - 1]) return;
    uint32_t k = atomicAdd(n_losers, 1u);
    losers[k] = sord[p];
}

struct LeafResolve {
    size_t n;
    const uint64_t* skey;      // [n] sorted first candidates
    const uint32_t* sord;      // [n] their owners
    uint32_t* heap;            // [n] min-heap of unresolved entities keyed by input order (starts as the losers, unsorted)
    uint32_t heap_n;
    uint8_t* evicted;          // [n] by sorted position: the holder lost its slot to an earlier entity
    uint64_t* tab_key;         // open-addressing set of slots taken by resolved colliders
    uint8_t* tab_used;
    uint32_t tab_mask;
};
__device__ inline void heap_push(uint32_t* h, uint32_t& n, uint32_t v) {
    uint32_t i = n++;
    h[i] = v;
    while (i > 0) {
        uint32_t p = (i - 1) >> 1;
        if (h[p] <= h[i]) break;
        uint32_t t = h[p]; h[p] = h[i]; h[i] = t;
        i = p;
    }
}
__device__ inline uint32_t heap_pop(uint32_t* h, uint32_t& n) {
    uint32_t top = h[0];
    h[0] = h[--n];
    uint32_t i = 0;
    for (;;) {
        uint32_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && h[l] < h[m]) m = l;
        if (r < n && h[r] < h[m]) m = r;
        if (m == i) break;
        uint32_t t = h[m]; h[m] = h[i]; h[i] = t;
        i = m;
    }
    return top;
}
// One lane, input order: every unresolved entity re-hashes until it finds a slot that no EARLIER entity holds;
// a later entity sitting on that slot is evicted and re-queued (it comes later in the order, so it is still ahead).
__global__ void k_leaf_resolve(LeafArgs A, LeafResolve R) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    uint32_t hn = 0;
    uint32_t n0 = R.heap_n;
    // heapify by successive pushes (the loser list arrives unsorted in R.heap)
    for (uint32_t i = 0; i < n0; i++) { uint32_t v = R.heap[i]; heap_push(R.heap, hn, v); }
    while (hn > 0) {
        uint32_t e = heap_pop(R.heap, hn);
        uint32_t st[8];
        for (int i = 0; i < 8; i++) st[i] = A.idx_state[(size_t)e * 8 + i];
        bool placed = false;
        for (int tries = 1; tries < A.max_tries && !placed; tries++) {      // the first try was the sorted candidate
            Digest d;
            dg_init(d, A.kind);
            dg_update_words(d, st, 8);
            dg_final(d, st);
            uint64_t x = leaf_index_of(st, A.height);
            // (a) slot taken by an already resolved collider (all of them precede e)
            bool taken = false;
            uint32_t hp = (uint32_t)((x * 0x9E3779B97F4A7C15ull) >> 32) & R.tab_mask;
            while (R.tab_used[hp]) {
                if (R.tab_key[hp] == x) { taken = true; break; }
                hp = (hp + 1) & R.tab_mask;
            }
            if (taken) continue;
            // (b) slot held by the first entity of a first-candidate group
            size_t lo = 0, hi = R.n;
            while (lo < hi) { size_t mid = (lo + hi) >> 1; if (R.skey[mid] < x) lo = mid + 1; else hi = mid; }
            if (lo < R.n && R.skey[lo] == x && !R.evicted[lo]) {
                uint32_t holder = R.sord[lo];
                if (holder < e) continue;                    // an earlier entity owns it
                R.evicted[lo] = 1;                           // e precedes the holder: e gets the slot, the holder re-queues
                heap_push(R.heap, hn, holder);
            }
            R.tab_used[hp] = 1;
            R.tab_key[hp] = x;
            A.cand[e] = x;
            placed = true;
        }
        if (!placed) { A.err[2] = 1; A.err[3] = e; return; }  // DapolError::FailedToMapIndex
    }
}

// ------------------------------------------------------------------------- parallel collision resolution
struct LeafTable {
    unsigned long long* key;   // [size] slot index, LEAF_EMPTY_KEY = free (indexes are < 2^63 here: height < 64)
    uint32_t* owner;           // [size] input position of the earliest entity that has claimed the slot (0xFFFFFFFF: none yet)
    uint32_t mask;             // size - 1
    uint32_t* pos;             // [n] table position of the entity's current candidate
    uint32_t* tries;           // [n] candidates drawn so far (1 = the first)
    uint32_t* flags;           // [0] somebody moved this round, [1] table full (fall back to the one-lane path)
};
#define LEAF_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull
// Claims slot x for entity e; returns the previous owner (> e or 0xFFFFFFFF: e owns it now; < e: an earlier entity does).
__device__ __forceinline__ uint32_t leaf_claim(const LeafTable& T, uint64_t x, uint32_t e, uint32_t& hp_out) {
    uint32_t hp = (uint32_t)((x * 0x9E3779B97F4A7C15ull) >> 32) & T.mask;
    for (uint32_t probes = 0; probes <= T.mask; probes++) {
        unsigned long long k = T.key[hp];
        if (k == LEAF_EMPTY_KEY) {
            k = atomicCAS(&T.key[hp], LEAF_EMPTY_KEY, (unsigned long long)x);
            if (k == LEAF_EMPTY_KEY) k = x;
        }
        if (k == x) { hp_out = hp; return atomicMin(&T.owner[hp], e); }
        hp = (hp + 1) & T.mask;
    }
    atomicOr(&T.flags[1], 1u);
    hp_out = 0;
    return 0;                                        // (reads as "lost"; the caller sees flags[1])
}
__global__ void k_leaf_claim_first(LeafArgs A, LeafTable T) {
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= A.n) return;
    uint32_t hp;
    (void)leaf_claim(T, A.cand[e], (uint32_t)e, hp);
    T.pos[e] = hp;
    T.tries[e] = 1;
}
// One round: every entity whose slot now belongs to an earlier entity draws candidates until it owns one (for now).
__global__ void k_leaf_settle(LeafArgs A, LeafTable T) {
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= A.n) return;
    uint32_t tries = T.tries[e];
    if (tries > (uint32_t)A.max_tries) return;       // parked: out of candidates in an earlier round
    if (T.owner[T.pos[e]] == (uint32_t)e) return;
    uint32_t st[8], hp = T.pos[e];
    for (int i = 0; i < 8; i++) st[i] = A.idx_state[e * 8 + i];
    uint64_t x = A.cand[e];
    for (;;) {
        if (tries >= (uint32_t)A.max_tries) {        // DapolError::FailedToMapIndex.  The entity parks (it holds no slot); the rounds go on
            atomicOr(&A.err[2], 1u);                 // to the fixed point, so that err[3] ends as the EARLIEST such entity -- the one the
            atomicMin(&A.err[3], (uint32_t)e);       // reference's loop stops at (the entities before it are where that loop puts them)
            tries = (uint32_t)A.max_tries + 1;
            break;
        }
        Digest d;
        dg_init(d, A.kind);
        dg_update_words(d, st, 8);
        dg_final(d, st);
        tries++;
        x = leaf_index_of(st, A.height);
        const uint32_t prev = leaf_claim(T, x, (uint32_t)e, hp);
        if (T.flags[1]) break;
        if (prev > (uint32_t)e) break;               // ours (until an earlier entity comes for it)
    }
    for (int i = 0; i < 8; i++) A.idx_state[e * 8 + i] = st[i];
    A.cand[e] = x;
    T.pos[e] = hp;
    T.tries[e] = tries;
    T.flags[0] = 1;
}

}  // namespace dapol
