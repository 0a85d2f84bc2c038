/* dapol_hip.h -- C ABI of the MI355X-native DAPOL+ proving path (libdapol_hip.so).
 *
 * The reference crate (MystenLabs/dapol, Rust) exposes no FFI; its seams are Rust traits and generic
 * parameters.  Each entry point below names the reference interface it replaces (paths relative to the
 * reference root).  A Rust maintainer binds these with an `extern "C"` block (INTEGRATION.md shows the stub).
 *
 * Conventions
 *  - every function returns an int32 status (DAPOL_OK == 0); nothing aborts or throws across the boundary;
 *  - all pointers are caller-owned HOST pointers unless the parameter name ends in `_dev` (device pointers,
 *    for callers that already keep their arrays in HBM);
 *  - byte strings are little-endian, 32-byte scalars / compressed ristretto255 points as in curve25519-dalek;
 *  - a dapol_ctx is bound to one GPU and one HIP stream; use one ctx per host thread / GPU.
 *
 * Randomness ("tape") contract: the reference draws from thread_rng() (src/dapol/node.rs:87 and inside
 * bulletproofs' prover), so its outputs differ run to run.  Here every draw is an explicit input:
 *  - a WIDE draw is 64 bytes reduced mod l (== Scalar::random);
 *  - seed mode: draw(domain, a, b) = first 64-byte XOF block of BLAKE3-keyed(seed, LE32 domain|LE64 a|LE64 b);
 *    padding node at (level above leaves, index): domain 1, a = level, b = index;
 *    range-proof nonce: domain 2, a = stream id (the leaf's tree index), b = slot, under a key bound to the STATEMENT:
 *    k1 = draw(seed; 6, stream id, first slot of the proof)[:32], k2 = draw(k1; 7, n_bits, m)[:32], then for every group
 *    of 31 value commitments k = BLAKE3(k | V_g .. V_g+30); the nonces are draw(k; 2, stream id, slot).  The crate draws
 *    fresh thread_rng randomness per call; a deterministic stream that ignored the statement would reuse a_blinding,
 *    s_L, s_R, tau against new challenges whenever a leaf is proved again after its siblings changed (dapol_tree_update,
 *    another policy / aggregation factor) and leak them.  Same statement => same nonces => same proof bytes;
 *  - slot order of one aggregated proof with m parties of n bits (the crate's draw order): party j draws
 *    a_blinding = j(2n+2), s_blinding = j(2n+2)+1, s_L[i] = j(2n+2)+2+i, s_R[i] = j(2n+2)+2+n+i; then
 *    t1_blinding_j = m(2n+2)+2j, t2_blinding_j = m(2n+2)+2j+1.   m(2n+4) slots in total;
 *  - tape mode: the same slots read from a caller buffer of 64-byte draws.
 */
#ifndef DAPOL_HIP_H
#define DAPOL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Status codes.  1-5 mirror DapolError (src/errors.rs:6-17); 6-7 mirror smtree DecodingError as used by
 * src/range/mod.rs:124-161 and src/proof/node.rs:81-102. */
enum {
    DAPOL_OK = 0,
    DAPOL_ERR_TREE_HEIGHT_TOO_BIG = 1,   /* DapolError::TreeHeightTooBig   (src/dapol/mod.rs:104-109) */
    DAPOL_ERR_SPARSITY_TOO_SMALL = 2,    /* DapolError::SparsityTooSmall   (src/dapol/mod.rs:110-116) */
    DAPOL_ERR_INVALID_DIGEST_SIZE = 3,   /* DapolError::InvalidDigestSize  (src/dapol/mod.rs:101-103) */
    DAPOL_ERR_DUPLICATED_INTERNAL_ID = 4,/* DapolError::DuplicatedInternalId (src/dapol/mod.rs:341-343) */
    DAPOL_ERR_FAILED_TO_MAP_INDEX = 5,   /* DapolError::FailedToMapIndex   (src/dapol/mod.rs:369-370) */
    DAPOL_ERR_BYTES_NOT_ENOUGH = 6,      /* DecodingError::BytesNotEnough */
    DAPOL_ERR_VALUE_DECODING = 7,        /* DecodingError::ValueDecodingError */
    DAPOL_ERR_INVALID_ARGUMENT = 8,      /* where the reference panics: unsorted/duplicate leaves (smtree build),
                                            aggregation_factor > #siblings (src/range/padding.rs:95-98), bad n/m
                                            (bulletproofs InvalidBitsize / InvalidAggregation) */
    DAPOL_ERR_UNKNOWN_LEAF = 9,          /* the `None` of Dapol::generate_proof* (src/dapol/mod.rs:148-190) */
    DAPOL_ERR_NO_DEVICE = 16,            /* no usable HIP device: the product path never falls back to the CPU */
    DAPOL_ERR_HIP = 17,                  /* a HIP runtime call failed; see dapol_last_error() */
    DAPOL_ERR_OUT_OF_MEMORY = 18,
    DAPOL_ERR_COMM = 19                  /* an RCCL call failed; see dapol_last_error() */
};

typedef struct dapol_ctx dapol_ctx;
typedef struct dapol_tree dapol_tree;

enum { DAPOL_POLICY_PADDING = 0, DAPOL_POLICY_SPLITTING = 1 };  /* RangeProofPadding / RangeProofSplitting */
/* D = blake3::Hasher (benches/dapol.rs:38), blake2::Blake2s (src/dapol/tests.rs:21), or blake2::Blake2b (src/tests.rs:100-101: the
 * reference's integration test also runs Dapol<Blake2b, R> through new_blank + build + prove + serialize + verify; only Dapol::new
 * rejects a digest that is not 32 bytes, src/dapol/mod.rs:101-103).  With DAPOL_DIGEST_BLAKE2B every node hash is 64 bytes: every
 * `H` buffer of this header -- whatever its parameter is called (H32, H_out32, path_H32 ...) -- then holds dapol_ctx_digest_bytes()
 * = 64 bytes per node instead of 32.  The Blake2b context serves the new_blank + build path: dapol_tree_build (enforce_sparsity = 0),
 * dapol_tree_update, roots / levels / paths, dapol_prove_entities, dapol_prove_batch, the wire format (*_d forms below),
 * dapol_verify_entities / dapol_verify_batch, dapol_merge_batch, dapol_padding_nodes, dapol_commit_hash_batch.  What the reference
 * refuses for such a digest is refused here with DAPOL_ERR_INVALID_DIGEST_SIZE: Dapol::new (dapol_tree_build with enforce_sparsity,
 * dapol_build_leaf_nodes); so are the paths that exist only for the benchmark and for multi-GPU sharding (workloads, shard trees,
 * the communicator, node records), whose records carry 32-byte hashes. */
enum { DAPOL_DIGEST_BLAKE3 = 0, DAPOL_DIGEST_BLAKE2S = 1, DAPOL_DIGEST_BLAKE2B = 2 };

/* Replaces the per-call PedersenGens::default() (src/dapol/node.rs:31, src/range/mod.rs:49,65,84,103) and
 * BulletproofGens::new(64, m) (src/range/mod.rs:50,66,85,104): generators and their window tables are derived
 * ONCE, on the GPU, for up to max_parties parties of 64 bits.  max_parties must be a power of two <= 1024.
 * digest_id: the node hash D of Dapol<D, R> used by every tree / merge / verify call of this context; anything but the
 * three digests above -> DAPOL_ERR_INVALID_DIGEST_SIZE. */
int32_t dapol_ctx_create(int32_t device, int32_t max_parties, int32_t digest_id, dapol_ctx** out);
int32_t dapol_ctx_digest_bytes(dapol_ctx* ctx, int32_t* bytes);          /* D::output_size(): 32 or 64 -- the size of every node hash of this context */
/* Settings of a context.  Every field: 0 = the library's own choice (what a plain dapol_ctx_create gives).  The first three are
 * fixed at creation (dapol_ctx_create_opts), the others may be changed at any time between calls (dapol_ctx_set_options).
 * None of them changes a byte of any output (tests/test_gpu_parity.py::test_every_proving_strategy_gives_the_same_bytes); they
 * trade memory against speed or pick among equivalent schedules.
 * The DAPOL_* ENVIRONMENT variables the sources mention are measurement knobs: they are read ONLY when the process has opted in
 * (DAPOL_ENV_KNOBS=1 in the environment, or dapol_env_knobs(1)) -- a host embedding this library does not inherit behaviour from
 * stray variables.  When enabled, a variable overrides the corresponding field. */
typedef struct {
    int32_t struct_size;              /* = sizeof(dapol_options): lets the struct grow */
    int32_t window_bits;              /* creation: width of the fixed-base windows, 8..20 (0: widest <= 17 whose tables fit table_gb) */
    double  table_gb;                 /* creation: budget of the window tables in GB (0: 40 GB, at most 30 % of the free memory) */
    int32_t high_half_rows;           /* creation: 0 auto, 1 on, -1 off: second table row per generator (halves the window steps of lanes that cannot share doublings) */
    int32_t generator_stationary;     /* 0 auto (calls of at least 1,024 proofs over >= 1,024 generators a side that are not small calls; batches of shorter proofs --
                                         a policy's individual proofs, 2 ... 8 parties -- from 2,048 / 1,024 / 512 proofs at 64 / 128 / 256-512 generators a side),
                                         1 every call that is not a small one, -1 never */
    int32_t gs_tile_rows;             /* rows per launch of the generator-stationary sweep (multiple of 4; 0: 16) */
    int32_t streams;                  /* chunks in flight, 1..4 (0: 1 -- one chunk of two rounds of resident wavefronts; more only share the chip) */
    int64_t chunk_proofs;             /* proofs per chunk (0: whole rounds of resident wavefronts, 65,536 on MI355X) */
    double  scratch_gb;               /* scratch budget of the range prover / verifier in GB (0: 130 GB, at most what is free beyond 8 GB) */
    int32_t tail_length;              /* length T of the hybrid inner-product argument's tail: 32 / 64 / 128 / 256, -1 = no tail (0: 64; swept batches of short
                                         proofs: 32 at 512 generators a side, none below) */
    int32_t small_call_max;           /* calls of up to this many proofs take the latency shapes.  0: 1,023 for proofs of >= 1,024 generators a side (the sweep
                                         takes over at 1,024 proofs unless generator_stationary = -1), 8,191 for smaller proofs -- where the short-list sweep
                                         takes over earlier (see generator_stationary; it yields only to generator_stationary = -1).  An explicit value is
                                         taken as it is for proofs of >= 1,024 generators a side */
    int32_t verify_batch_min;         /* fewest proofs the verifier checks as ONE random linear combination (0: 112) */
    int64_t update_incremental_max;   /* dapol_tree_update: most replaced leaves re-merged in place (0: 65,536; -1: always rebuild) */
    int32_t gs_slices;                /* slices of a list swept side by side by the generator-stationary MSM: 1, 2, 4, 8, 16 (0: as many as fill the chip) */
    int32_t profile;                  /* creation: DAPOL_PROFILE_BENCH (0, the default) or DAPOL_PROFILE_HOST: the defaults of the memory fields above.
                                         BENCH spends HBM for the last percent of throughput: 17-bit windows + high-half rows (69 GB of tables for 32
                                         parties), chunks of two rounds of resident wavefronts (102 GB of scratch during a large call).  HOST is sized for
                                         an embedder that shares the GPU: 18 GB table budget (16-bit windows for 32 parties, no high-half rows), chunks of
                                         one round (<= 56 GB of scratch) -- INTEGRATION.md section 4 gives both footprints and throughputs.  Explicit
                                         table_gb / window_bits / high_half_rows / scratch_gb / chunk_proofs win over the profile. */
} dapol_options;
enum { DAPOL_PROFILE_BENCH = 0, DAPOL_PROFILE_HOST = 1 };
int32_t dapol_ctx_create_opts(int32_t device, int32_t max_parties, int32_t digest_id, const dapol_options* options, dapol_ctx** out);
int32_t dapol_ctx_get_options(dapol_ctx* ctx, dapol_options* out);      /* the stored settings (zeros = defaults) + window_bits / high_half_rows as built */
int32_t dapol_ctx_set_options(dapol_ctx* ctx, const dapol_options* options);   /* creation-time fields are ignored here */
int32_t dapol_env_knobs(int32_t enable);                                 /* process-wide opt-in to the DAPOL_* measurement knobs; returns the old setting */
/* Trees and workloads built on a context keep it alive: dapol_ctx_destroy releases the caller's handle, the storage goes
 * when the last tree / workload of the context is destroyed too -- handles may be destroyed in any order. */
int32_t dapol_ctx_destroy(dapol_ctx* ctx);
/* Compressed generators (for cross-checks): which = 0 B, 1 B_blinding, 2 G[party][bit], 3 H[party][bit]. */
int32_t dapol_ctx_generator(dapol_ctx* ctx, int32_t which, int32_t party, int32_t bit, uint8_t out32[32]);
const char* dapol_strerror(int32_t code);
const char* dapol_last_error(void);
/* Diagnostics: how many times a call was left between a fork onto one of the context's side streams and the matching join (an error
 * return) and therefore waited for that stream before returning, process-wide.  0 in a healthy run. */
int32_t dapol_diag_fork_guard_waits(uint64_t* count);
/* Diagnostics: combined (random-linear-combination) checks of dapol_range_verify_batch / dapol_verify_* that failed and went on to
 * bisection or the proof-by-proof check, process-wide.  Stays 0 while every proof handed in is valid. */
int32_t dapol_diag_verify_fallbacks(uint64_t* count);
/* Diagnostics: device milliseconds of the proving pipeline of the last dapol_range_prove_batch in this process (HIP events on the
 * context's stream: inputs already in HBM, proofs not yet copied back). */
int32_t dapol_diag_range_prove_ms(double* ms);

/* DapolNode::new (src/dapol/node.rs:29-45), batched: C_i = v_i*B + r_i*B_blinding (compressed), H_i = D(C_i).
 * r may be an unreduced Scalar::from_bits value (bit 255 clear; src/dapol/mod.rs:385). */
int32_t dapol_commit_hash_batch(dapol_ctx* ctx, size_t n, const uint64_t* v, const uint8_t* r32, uint8_t* C_out32,
                                uint8_t* H_out32);

/* build_leaf_nodes + shuffle_index (src/dapol/mod.rs:323-441), the leaf half of Dapol::new: liabilities in INPUT order
 * (ids as concatenated bytes with n+1 offsets) -> for every liability its tree index and blinding factor, returned both
 * per input entity (idx_by_entity, may be NULL = the id_to_idx_map of mod.rs:388) and sorted by index, ready for
 * dapol_tree_build (mod.rs:396): leaf_idx_sorted / v_sorted / r32_sorted, with order_sorted[p] = input position of the
 * p-th leaf.  digest_id: DAPOL_DIGEST_BLAKE3 or DAPOL_DIGEST_BLAKE2S (the digest of the reference's known-answer tests,
 * src/dapol/tests.rs:13,21); ids of any length, as in the reference (mod.rs:347-349, 358-360) -- BLAKE3 inputs beyond one
 * 1024-byte chunk go through the tree mode on the device.  Collisions are resolved exactly as the
 * reference does (an entity competes only with the entities before it in the input; up to 128 re-hashes).
 * Errors: DAPOL_ERR_TREE_HEIGHT_TOO_BIG, DAPOL_ERR_SPARSITY_TOO_SMALL (2^height < 2n), DAPOL_ERR_DUPLICATED_INTERNAL_ID,
 * DAPOL_ERR_FAILED_TO_MAP_INDEX. */
int32_t dapol_build_leaf_nodes(dapol_ctx* ctx, int32_t digest_id, const uint8_t* audit_seed, size_t audit_seed_len, int32_t height,
                               size_t n, const uint8_t* internal_ids, const uint32_t* internal_off, const uint8_t* external_ids,
                               const uint32_t* external_off, const uint64_t* values, uint64_t* leaf_idx_sorted, uint64_t* v_sorted,
                               uint8_t* r32_sorted, uint32_t* order_sorted, uint64_t* idx_by_entity);

/* Dapol::new_blank + Dapol::build (src/dapol/mod.rs:196-208 -> smtree SparseMerkleTree::build) with the leaf
 * nodes made by DapolNode::new; also the build half of Dapol::new (src/dapol/mod.rs:100-128).
 * leaf_idx must be strictly increasing and < 2^height (the reference panics otherwise -> INVALID_ARGUMENT);
 * enforce_sparsity != 0 applies the 2^height >= 2n check of Dapol::new.  Padding nodes (node.rs:86-88) take
 * their blinding from pad_seed32 (seed mode, positional). */
int32_t dapol_tree_build(dapol_ctx* ctx, int32_t height, size_t n, const uint64_t* leaf_idx, const uint64_t* v,
                         const uint8_t* r32, const uint8_t pad_seed32[32], int32_t enforce_sparsity, dapol_tree** out);
/* TAPE mode of the build (the randomness contract in this header's head comment): the padding nodes' blindings are read from
 * `tape` -- tape_draws x 64 bytes, each reduced mod l like Scalar::random, ONE PER PADDING NODE in the order (level bottom-up, index
 * ascending) -- instead of being derived from a seed.  dapol_tree_padding_positions (host only, pure index arithmetic) gives that
 * order and the count for a leaf set: *count, and, when non-NULL, level[] (0 = leaf level) and index[] of every padding node.  With
 * the draws a seed would have given, the tree equals dapol_tree_build's bit for bit.  A tape shorter than the tree's padding nodes ->
 * DAPOL_ERR_INVALID_ARGUMENT.  A tree built from a tape cannot be updated (no seed to draw new padding nodes from). */
int32_t dapol_tree_padding_positions(int32_t height, size_t n, const uint64_t* leaf_idx, size_t* count, uint8_t* level, uint64_t* index);
int32_t dapol_tree_build_tape(dapol_ctx* ctx, int32_t height, size_t n, const uint64_t* leaf_idx, const uint64_t* v, const uint8_t* r32,
                              const uint8_t* tape, size_t tape_draws, dapol_tree** out);
/* Multi-GPU sharding: builds only the subtree that holds all the given leaves, i.e. the
 * lowest (total_height - shard_bits) levels; every leaf index (GLOBAL, < 2^total_height) must share its top
 * shard_bits bits.  Padding seeds and nonce stream ids use the global indexes, so the nodes equal the ones a
 * single-GPU build of the whole tree produces.  dapol_tree_root then returns the subtree root. */
int32_t dapol_tree_build_shard(dapol_ctx* ctx, int32_t total_height, int32_t shard_bits, size_t n, const uint64_t* leaf_idx,
                               const uint64_t* v, const uint8_t* r32, const uint8_t pad_seed32[32], dapol_tree** out);
/* Paddable::padding (src/dapol/node.rs:86-88), batched: the padding node new(0, blinding) at (level above the leaves,
 * index within that level), its blinding drawn positionally from pad_seed (domain 1 of the randomness contract) -- the
 * very node dapol_tree_build puts there.  Value is 0.  Used e.g. as the record of an empty shard in a multi-GPU build. */
int32_t dapol_padding_nodes(dapol_ctx* ctx, const uint8_t pad_seed32[32], size_t n, const uint8_t* level, const uint64_t* index,
                            uint8_t* C32, uint8_t* H32, uint8_t* r32);
/* Mergeable::merge (src/dapol/node.rs:64-80; the (C,H) half is DapolProofNode::merge, src/proof/node.rs:56-69),
 * batched on compressed inputs: parent[i] = merge(left[i], right[i]).  Used for the replicated top levels above
 * the shard roots and by the Merkle-path re-merge of verification.  v/r pointers may all be NULL ((C,H) only).
 * A commitment that does not decode -> DAPOL_ERR_VALUE_DECODING. */
int32_t dapol_merge_batch(dapol_ctx* ctx, size_t n, const uint8_t* CL32, const uint8_t* HL32, const uint64_t* vL, const uint8_t* rL32,
                          const uint8_t* CR32, const uint8_t* HR32, const uint64_t* vR, const uint8_t* rR32, uint8_t* C32,
                          uint8_t* H32, uint64_t* v, uint8_t* r32);
int32_t dapol_tree_destroy(dapol_tree* tree);
/* Dapol::update (src/dapol/mod.rs:211-213 -> SparseMerkleTree::update), batched: the k leaves are applied in input
 * order -- a leaf is inserted, or replaces the liability already at its index (the last of several updates of one
 * index wins).  Afterwards the tree equals dapol_tree_build of the resulting leaf set bit for bit (padding nodes
 * are keyed by position, so the reference test's build-vs-update root equality, src/tests.rs:48, holds exactly).
 * Only for trees from dapol_tree_build / dapol_tree_build_shard.  An error reported before anything was written (bad arguments,
 * an index outside the tree or the shard, an allocation that failed) leaves the tree unchanged.  Small batches are re-merged IN
 * PLACE on the device; a HIP failure between the first and the last write of such an update leaves upper levels that no longer
 * match the leaves: the tree is then marked invalid and every later call on it returns DAPOL_ERR_INVALID_ARGUMENT ("left
 * inconsistent ...") -- destroy it and build it again.  The same holds for a batch that mixes new and existing indexes once its
 * inserts have been applied: whatever stops the replacements after that -- also an error before THEIR first write -- would leave a
 * consistent but half-updated tree, so the tree is marked invalid too (never "unchanged" with half the batch in).
 * On a context with a 64-byte digest the whole hash chain is laid again after the update; a tree built from a padding tape
 * (dapol_tree_build_tape) cannot be updated. */
int32_t dapol_tree_update(dapol_tree* tree, size_t k, const uint64_t* leaf_idx, const uint64_t* v, const uint8_t* r32);
/* What the last dapol_tree_update on this tree did: 0 = rebuilt, 1 = replaced existing leaves in place, 2 = inserted new leaves in
 * place, 3 = both (diagnostics; the tree is the same whichever path ran). */
int32_t dapol_tree_last_update_path(dapol_tree* tree, int32_t* path);
/* Dapol::root_raw / Dapol::root (src/dapol/mod.rs:134-141). Any out pointer may be NULL. */
int32_t dapol_tree_root(dapol_tree* tree, uint8_t C32[32], uint8_t H32[32], uint64_t* v, uint8_t r32[32]);
int32_t dapol_tree_node_count(dapol_tree* tree, uint64_t* real_nodes, uint64_t* padding_nodes);
/* Every stored node of one level (0 = leaves .. height = root), real nodes first then padding nodes, for
 * parity tests on small trees.  Arrays sized by dapol_tree_level_size. */
int32_t dapol_tree_level_size(dapol_tree* tree, int32_t level, uint64_t* n_real, uint64_t* n_pad);
int32_t dapol_tree_level_nodes(dapol_tree* tree, int32_t level, uint64_t* idx, uint64_t* v, uint8_t* r32, uint8_t* C32,
                               uint8_t* H32, uint8_t* is_pad);
/* The sibling half of Dapol::generate_proof_batch for single leaves (src/dapol/mod.rs:172-184): for each of the
 * b leaves, its `height` siblings root side first: (C, H) = the Merkle path proof nodes, (v, r) = the secrets
 * handed to R::generate_proof.  Unknown leaf -> DAPOL_ERR_UNKNOWN_LEAF.  Out pointers may be NULL. */
int32_t dapol_tree_paths(dapol_tree* tree, size_t b, const uint64_t* leaf_idx, uint8_t* sib_C32, uint8_t* sib_H32,
                         uint64_t* sib_v, uint8_t* sib_r32);

/* generate_aggregated_range_proof / generate_single_range_proof (src/range/mod.rs:48-78 ->
 * RangeProof::prove_multiple / prove_single), batched over b independent proofs of m parties x n_bits each
 * (reference: n_bits = 64, src/range/mod.rs:16).  v[b][m], r32[b][m][32];  proofs_out[b][32*(9+2*lg(n_bits*m))].
 * Nonces: seed mode (nonce_seed32, stream_id[b], slot_base) or tape mode (tape != NULL:
 * tape[b][m*(2*n_bits+4)][64], slot_base ignored). */
int32_t dapol_range_prove_batch(dapol_ctx* ctx, int32_t n_bits, int32_t m, size_t b, const uint64_t* v, const uint8_t* r32,
                                const uint8_t nonce_seed32[32], const uint64_t* stream_id, uint64_t slot_base,
                                const uint8_t* tape, uint8_t* proofs_out);
size_t dapol_range_proof_size(int32_t n_bits, int32_t m);

/* verify_aggregated_range_proof / verify_single_range_proof (src/range/mod.rs:83-119 -> verify_multiple), batched:
 * proofs[b][size], V32[b][m][32] -> ok[b] (1 = verifies).
 * verify_seed32 (here and in dapol_verify_entities / dapol_verify_batch) is SOUNDNESS-CRITICAL: it keys the verifier's
 * random scalars -- c, which combines the two equations of one proof (thread_rng in the crate), and the weights of the
 * cross-proof batch check.  Pass NULL (recommended): the library draws 32 bytes from the OS CSPRNG per call.  A caller
 * seed is for reproducible tests; even then every scalar is derived from BLAKE3(seed | digest of every proof and
 * commitment of the batch), so a weight cannot be predicted before the proofs are fixed (Fiat-Shamir). */
int32_t dapol_range_verify_batch(dapol_ctx* ctx, int32_t n_bits, int32_t m, size_t b, const uint8_t* proofs, const uint8_t* V32,
                                 const uint8_t verify_seed32[32], uint8_t* ok);

/* Dapol::generate_proof for b single leaves (src/dapol/mod.rs:167-190 + R::generate_proof,
 * src/range/padding.rs:88-118 / src/range/splitting.rs:100-129): gathers each leaf's siblings on the GPU and
 * proves them under `policy` with `aggregation_factor`.  Outputs, per leaf: the Merkle path proof nodes
 * (path_C32/path_H32: [b][height][32], root side first; may be NULL) and the range proofs concatenated in
 * generation order (aggregated proofs, then the individual 672-byte proofs): range_out[b][dapol_entity_proof_size].
 * stream id of a leaf = its tree index. */
int32_t dapol_prove_entities(dapol_ctx* ctx, dapol_tree* tree, size_t b, const uint64_t* leaf_idx, int32_t policy,
                             int32_t aggregation_factor, int32_t n_bits, const uint8_t nonce_seed32[32], uint8_t* path_C32,
                             uint8_t* path_H32, uint8_t* range_out);
size_t dapol_entity_proof_size(int32_t height, int32_t policy, int32_t aggregation_factor, int32_t n_bits);
/* TAPE mode of dapol_prove_entities: the nonces of entity e are the draws tape[e][slot][64], slot < dapol_entity_tape_slots(...), in
 * the crate's draw order -- the sub-proofs of the policy one after the other (ONE RNG runs through R::generate_proof,
 * src/range/padding.rs:104-112, splitting.rs:110-123), inside each the slot order given at the top of this header.  What a Rust
 * harness replays through a custom RngCore into prove_multiple_with_rng (tools/replay_tape.rs). */
size_t dapol_entity_tape_slots(int32_t height, int32_t policy, int32_t aggregation_factor, int32_t n_bits);
int32_t dapol_prove_entities_tape(dapol_ctx* ctx, dapol_tree* tree, size_t b, const uint64_t* leaf_idx, int32_t policy, int32_t aggregation_factor,
                                  int32_t n_bits, const uint8_t* tape, uint8_t* path_C32, uint8_t* path_H32, uint8_t* range_out);
/* Same, for a shard tree: the n_upper siblings above the shard root (root side first; identical for every leaf
 * of the shard) are prepended to each leaf's own siblings before the policy is applied.  Path outputs are
 * [b][n_upper + tree height][32]. */
int32_t dapol_prove_entities_upper(dapol_ctx* ctx, dapol_tree* tree, size_t b, const uint64_t* leaf_idx, int32_t policy,
                                   int32_t aggregation_factor, int32_t n_bits, const uint8_t nonce_seed32[32], int32_t n_upper,
                                   const uint8_t* up_C32, const uint8_t* up_H32, const uint64_t* up_v, const uint8_t* up_r32,
                                   uint8_t* path_C32, uint8_t* path_H32, uint8_t* range_out);

/* Serializable for RangeProofPadding / RangeProofSplitting (src/range/padding.rs:38-69, src/range/splitting.rs:36-84):
 * the range-proof blob of ONE entity as written by dapol_prove_entities <-> R::serialize() bytes
 * ((aggregated_num ||) (size || proof)... || individual_num || proofs...; field widths src/range/mod.rs:18-21).
 * Host-only.  Deserialisation applies RangeProof::from_bytes' framing / canonical-scalar checks and returns
 * DAPOL_ERR_BYTES_NOT_ENOUGH / DAPOL_ERR_VALUE_DECODING like deserialize_range_proof (src/range/mod.rs:124-139). */
size_t dapol_range_proofs_wire_size(int32_t height, int32_t policy, int32_t aggregation_factor, int32_t n_bits);
int32_t dapol_range_proofs_serialize(int32_t height, int32_t policy, int32_t aggregation_factor, int32_t n_bits, const uint8_t* blob,
                                     uint8_t* wire_out);
int32_t dapol_range_proofs_deserialize(int32_t policy, int32_t n_bits, const uint8_t* wire, size_t wire_len, uint8_t* blob_out, size_t blob_cap,
                                       uint32_t* n_aggregated, uint64_t* agg_sizes, uint64_t* n_individual, size_t* consumed);

/* What smtree 0.1.2 owns and nothing in the reference repository pins (the crate is not vendored, no golden bytes exist):
 * one field per assumption, the defaults being the believed values.  Process-wide; set it before proving / serialising.
 * A maintainer with cargo closes these with tools/replay_tape.rs -> tests/golden/from_reference/ (INTEGRATION.md). */
typedef struct {
    int32_t int_big_endian;       /* smtree::utils::usize_to_bytes byte order: 1 = big-endian (default), 0 = little-endian */
    int32_t batch_num_bytes;      /* MerkleProof::serialize: width of the leaf-count field (default 8) */
    int32_t sibling_num_bytes;    /* MerkleProof::serialize: width of the sibling-count field (default 8) */
    int32_t tree_height_bytes;    /* TreeIndex::serialize: width of the height field (default 2) */
    int32_t path_bytes_full;      /* 0 (default): a path takes ceil(height / 8) bytes; 1: all 32 bytes of TreeIndex.pos */
    int32_t siblings_leaf_first;  /* order of a proof's siblings = of the range proof's parties: 0 (default) from the root side
                                     down, 1 from the leaf level up (batched proofs: level by level in that direction) */
} dapol_wire_config;
int32_t dapol_wire_config_get(dapol_wire_config* out);
int32_t dapol_wire_config_set(const dapol_wire_config* cfg);

/* Serializable for DapolProofNode (src/proof/node.rs:74-102), n nodes: wire = (C32 || hash32) per node.  Deserialisation
 * decompress-validates every commitment in one GPU launch: DAPOL_ERR_BYTES_NOT_ENOUGH / DAPOL_ERR_VALUE_DECODING ("Not the
 * canonical encoding of a point."). */
int32_t dapol_proof_nodes_serialize(size_t n, const uint8_t* C32, const uint8_t* H32, uint8_t* wire_out);
int32_t dapol_proof_nodes_deserialize(dapol_ctx* ctx, size_t n, const uint8_t* wire, size_t wire_len, uint8_t* C32_out, uint8_t* H32_out);
/* ... for a digest of hash_bytes = D::output_size() bytes (32 or 64): wire = (C32 || hash) per node, H holds hash_bytes per node.
 * (The context-taking deserialisers read the width from their context.) */
int32_t dapol_proof_nodes_serialize_d(int32_t hash_bytes, size_t n, const uint8_t* C32, const uint8_t* H, uint8_t* wire_out);

/* DapolProof::serialize / deserialize (src/proof/mod.rs:68-85): R::serialize() || MerkleProof::serialize(), the latter
 * restated as batch_num || sibling_num || tree_height || path_1..k || (C || hash)_1..S (smtree 0.1.2, from memory: see
 * dapol_wire_config).  k leaves (k = 1: a proof of dapol_prove_entities with n_siblings = height and the path arrays as
 * siblings; k > 1: the outputs of dapol_prove_batch); range_blob = the range proofs in dapol_prove_entities' blob layout.
 * dapol_proof_deserialize: first call with all four output arrays NULL -> height, k, n_siblings, aggregation_factor
 * (= n_siblings - number of individual proofs, src/range/padding.rs:172) and range_blob_len; second call fills the
 * arrays.  Errors as the reference: DAPOL_ERR_BYTES_NOT_ENOUGH, DAPOL_ERR_VALUE_DECODING (range-proof framing, non-canonical
 * scalars, sibling commitments that do not decompress -- validated on the GPU). */
size_t dapol_proof_wire_size(int32_t height, size_t k, size_t n_siblings, int32_t policy, int32_t aggregation_factor, int32_t n_bits);
int32_t dapol_proof_serialize(int32_t height, size_t k, const uint64_t* leaf_idx, size_t n_siblings, const uint8_t* sib_C32,
                              const uint8_t* sib_H32, int32_t policy, int32_t aggregation_factor, int32_t n_bits,
                              const uint8_t* range_blob, uint8_t* wire_out);
size_t dapol_proof_wire_size_d(int32_t hash_bytes, int32_t height, size_t k, size_t n_siblings, int32_t policy, int32_t aggregation_factor, int32_t n_bits);
int32_t dapol_proof_serialize_d(int32_t hash_bytes, int32_t height, size_t k, const uint64_t* leaf_idx, size_t n_siblings, const uint8_t* sib_C32,
                                const uint8_t* sib_H, int32_t policy, int32_t aggregation_factor, int32_t n_bits, const uint8_t* range_blob,
                                uint8_t* wire_out);
int32_t dapol_proof_deserialize(dapol_ctx* ctx, int32_t policy, int32_t n_bits, const uint8_t* wire, size_t wire_len, int32_t* height,
                                size_t* k, size_t* n_siblings, int32_t* aggregation_factor, size_t* range_blob_len,
                                uint64_t* leaf_idx_out, uint8_t* sib_C32_out, uint8_t* sib_H32_out, uint8_t* range_blob_out,
                                size_t* consumed);

/* DapolProof::verify (src/proof/mod.rs:41-47 + :89-95) for b single-leaf inclusion proofs as produced by
 * dapol_prove_entities: MerkleProof::verify by re-merging the leaf proof node with its siblings (DapolProofNode::merge,
 * src/proof/node.rs:56-69; a sibling commitment that is not a canonical point fails like deserialisation does, :88-94)
 * and then RangeVerifiable::verify of the policy over the sibling commitments.  ok[i] = 1 iff both hold. */
int32_t dapol_verify_entities(dapol_ctx* ctx, int32_t height, size_t b, const uint64_t* leaf_idx, const uint8_t* leaf_C32,
                              const uint8_t* leaf_H32, const uint8_t* path_C32, const uint8_t* path_H32, const uint8_t root_C32[32],
                              const uint8_t root_H32[32], int32_t policy, int32_t aggregation_factor, int32_t n_bits,
                              const uint8_t* range_proofs, const uint8_t verify_seed32[32], uint8_t* ok);

/* Batched inclusion proofs -- Dapol::generate_proof_batch for k >= 1 leaves (src/dapol/mod.rs:172-190) and
 * DapolProof::verify_batch (src/proof/mod.rs:49-54): ONE proof for all k leaves = the deduplicated siblings of the
 * batched Merkle paths + one R::generate_proof over exactly those siblings (values / blindings in sibling order).
 * leaf_idx must be strictly increasing.  Sibling order: level by level from the root side, left to right within a
 * level (for k = 1 this is dapol_tree_paths' order; smtree's own order is not pinned by the reference repository).
 *
 * dapol_batch_siblings: number of siblings and, optionally, their positions (sib_level: 0 = leaf level; sib_index:
 * node index within its level).  Pure index arithmetic -- no GPU, no tree.
 * dapol_prove_batch: sib_C32 / sib_H32: [n_siblings][32] (may be NULL); range_out:
 * dapol_entity_proof_size(n_siblings, policy, aggregation_factor, n_bits) bytes in dapol_prove_entities' blob layout.
 * aggregation_factor > n_siblings -> DAPOL_ERR_INVALID_ARGUMENT (the reference panics, src/range/padding.rs:95-98).
 * For k > 1 the nonce stream is keyed by the seed chained through the leaf list, so a batch never shares nonces with
 * a single-leaf proof made from the same seed; k = 1 gives exactly dapol_prove_entities' bytes.
 * dapol_verify_batch: *ok = 1 iff the leaves and siblings re-merge to the root and R::verify accepts. */
int32_t dapol_batch_siblings(int32_t height, size_t k, const uint64_t* leaf_idx, size_t* n_siblings, uint8_t* sib_level,
                             uint64_t* sib_index);
int32_t dapol_prove_batch(dapol_ctx* ctx, dapol_tree* tree, size_t k, const uint64_t* leaf_idx, int32_t policy,
                          int32_t aggregation_factor, int32_t n_bits, const uint8_t nonce_seed32[32], uint8_t* sib_C32,
                          uint8_t* sib_H32, uint8_t* range_out);
int32_t dapol_verify_batch(dapol_ctx* ctx, int32_t height, size_t k, const uint64_t* leaf_idx, const uint8_t* leaf_C32,
                           const uint8_t* leaf_H32, size_t n_siblings, const uint8_t* sib_C32, const uint8_t* sib_H32,
                           const uint8_t root_C32[32], const uint8_t root_H32[32], int32_t policy, int32_t aggregation_factor,
                           int32_t n_bits, const uint8_t* range_proofs, const uint8_t verify_seed32[32], uint8_t* ok);

/* Length-checked forms for buffers that come from UNTRUSTED bytes (a decoded wire): the counts the caller holds are compared
 * with what (height, policy, aggregation_factor, n_bits) imply before anything is read; a proof of the wrong shape is an INVALID
 * proof (ok = 0, DAPOL_OK), never an over-read.  n_path_nodes = records in path_C32 / path_H32 (must be b * height);
 * range_proofs_len in bytes (must be b * dapol_entity_proof_size(height, ...), or the size over n_siblings for a batch proof). */
int32_t dapol_verify_entities_checked(dapol_ctx* ctx, int32_t height, size_t b, const uint64_t* leaf_idx, const uint8_t* leaf_C32, const uint8_t* leaf_H32,
                                      size_t n_path_nodes, const uint8_t* path_C32, const uint8_t* path_H32, const uint8_t root_C32[32],
                                      const uint8_t root_H32[32], int32_t policy, int32_t aggregation_factor, int32_t n_bits, const uint8_t* range_proofs,
                                      size_t range_proofs_len, const uint8_t verify_seed32[32], uint8_t* ok);
int32_t dapol_verify_batch_checked(dapol_ctx* ctx, int32_t height, size_t k, const uint64_t* leaf_idx, const uint8_t* leaf_C32, const uint8_t* leaf_H32,
                                   size_t n_siblings, const uint8_t* sib_C32, const uint8_t* sib_H32, const uint8_t root_C32[32], const uint8_t root_H32[32],
                                   int32_t policy, int32_t aggregation_factor, int32_t n_bits, const uint8_t* range_proofs, size_t range_proofs_len,
                                   const uint8_t verify_seed32[32], uint8_t* ok);

/* Multi-GPU: one process per GPU, rank g owning the top-level subtree with index prefix g (dapol_tree_build_shard /
 * dapol_workload_create_shard).  The reference has no communication (single process, single thread); the sharded path has
 * exactly one exchange step and one final reduce, both RCCL calls inside this library (librccl.so, over xGMI):
 *   dapol_comm_unique_id   rank 0 makes the 128-byte RCCL id; the host distributes it by whatever means it has
 *   dapol_comm_create      the communicator on the context's GPU (world must be a power of two), created NON-BLOCKING
 *                          (ncclCommInitRankConfig, blocking = 0) and polled with ncclCommGetAsyncError.
 *   dapol_comm_create_timeout  ... up to a deadline: a communicator that has not come up after timeout_ms is aborted
 *                          (ncclCommAbort) and the call returns DAPOL_ERR_COMM, so that one absent rank cannot hang the others
 *                          inside ncclCommInitRank for good; timeout_ms <= 0 waits without a deadline.  The communicator keeps
 *                          the deadline for its COLLECTIVES: a dapol_shard_exchange / dapol_comm_allreduce_u64 that has not
 *                          completed after timeout_ms, or during which RCCL reports an asynchronous error, aborts the
 *                          communicator (which ends an RCCL kernel waiting for a peer), returns DAPOL_ERR_COMM, and every
 *                          later call on the handle fails at once -- the host falls back to another transport or exits.
 *   dapol_comm_abort       the failure path (ncclCommAbort): tear down without waiting for the peers, once the ranks have
 *                          agreed not to use this communicator; dapol_comm_destroy is the orderly end (finalize + destroy).
 *   dapol_comm_count       the number of ranks as RCCL reports it (ncclCommCount)
 *   dapol_shard_exchange   ncclAllGather of the G subtree-root records (C | H | v LE64 | r = 104 bytes each; an empty shard
 *                          sends the padding node of its root position, dapol_padding_nodes), then every rank merges the
 *                          log2 G replicated top levels on its own GPU (Mergeable::merge) -> the global root record and the
 *                          upper siblings of this rank, root side first, ready for dapol_prove_entities_upper /
 *                          dapol_workload_prove.  records_out (may be NULL): the gathered [world][104] records.
 *   dapol_comm_allreduce_u64  ncclAllReduce over 64-bit words: the final reduce of the per-rank proof checksums (wrapping
 *                          sum), proof counts, or verdict masks (min = AND of 0/1 verdicts).
 *   dapol_shard_top_levels the merge half alone, for records gathered by other means (tests, other transports). */
typedef struct dapol_comm dapol_comm;
enum { DAPOL_COMM_ID_BYTES = 128, DAPOL_RECORD_BYTES = 104 };
enum { DAPOL_REDUCE_SUM = 0, DAPOL_REDUCE_MIN = 1, DAPOL_REDUCE_MAX = 2 };
int32_t dapol_comm_unique_id(uint8_t id_out[DAPOL_COMM_ID_BYTES]);
int32_t dapol_comm_create(dapol_ctx* ctx, const uint8_t id[DAPOL_COMM_ID_BYTES], int32_t rank, int32_t world, dapol_comm** out);
int32_t dapol_comm_create_timeout(dapol_ctx* ctx, const uint8_t id[DAPOL_COMM_ID_BYTES], int32_t rank, int32_t world, int64_t timeout_ms, dapol_comm** out);
int32_t dapol_comm_destroy(dapol_comm* comm);
int32_t dapol_comm_abort(dapol_comm* comm);
int32_t dapol_comm_count(dapol_comm* comm, int32_t* count);
int32_t dapol_shard_exchange(dapol_comm* comm, const uint8_t sub_C[32], const uint8_t sub_H[32], uint64_t sub_v, const uint8_t sub_r[32],
                             uint8_t root_C[32], uint8_t root_H[32], uint64_t* root_v, uint8_t root_r[32], uint8_t* up_C32, uint8_t* up_H32,
                             uint64_t* up_v, uint8_t* up_r32, uint8_t* records_out);
int32_t dapol_shard_top_levels(dapol_ctx* ctx, int32_t world, int32_t rank, const uint8_t* records, uint8_t root_C[32], uint8_t root_H[32],
                               uint64_t* root_v, uint8_t root_r[32], uint8_t* up_C32, uint8_t* up_H32, uint64_t* up_v, uint8_t* up_r32);
int32_t dapol_comm_allreduce_u64(dapol_comm* comm, int32_t op, uint64_t* inout, size_t n);
/* What the two collectives cost, measured inside the calls above (a multi-GPU bench line carries them per step, so that a scaling
 * curve can be read from its own records): DEVICE time between HIP events on the context's stream around ncclAllGather -- which
 * includes waiting for the slowest rank to arrive -- around the replicated merge of the top levels with its copies back, and
 * around ncclAllReduce; HOST wall time of each whole call.  Microseconds; last_* = the most recent call, sum_* since creation or
 * the last reset (reset != 0 clears the counters after copying them out). */
typedef struct {
    uint64_t exchanges, reduces;
    double last_allgather_us, last_top_levels_us, last_exchange_host_us, last_allreduce_us, last_reduce_host_us;
    double sum_allgather_us, sum_top_levels_us, sum_exchange_host_us, sum_allreduce_us, sum_reduce_host_us;
} dapol_comm_timing;
int32_t dapol_comm_timing_get(dapol_comm* comm, dapol_comm_timing* out, int32_t reset);

/* generate_proof_batch (src/dapol/mod.rs:172-190) on a SHARDED tree.  The siblings of a batch lie in several shards and in
 * the replicated top levels, so the proof is assembled from node RECORDS (C, H, v, r):
 *   dapol_batch_siblings            the positions (level above the leaves, index) of the proof's siblings;
 *   dapol_tree_node_records         this rank's records among them: the real or padding node a (shard) tree stores at each
 *                                   position, found[i] = 0 where the tree holds none (another shard's, or above the shard root);
 *   dapol_shard_top_node_records    the nodes at and above the shard roots (level_above = 0 .. log2 G), from the G exchanged
 *                                   subtree-root records;
 *   dapol_prove_batch_records       R::generate_proof over the assembled siblings, in dapol_batch_siblings' order.
 * Between the second and the last step the ranks exchange their 104-byte rows by any transport.  The result equals
 * dapol_prove_batch on the unsharded tree byte for byte.  dapol_workload_tree lends a workload's current (sub)tree to these
 * calls (valid until the workload's next build / destroy; do not destroy it). */
int32_t dapol_tree_node_records(dapol_tree* tree, size_t n, const uint8_t* level, const uint64_t* index, uint8_t* C32, uint8_t* H32,
                                uint64_t* v, uint8_t* r32, uint8_t* found);
int32_t dapol_shard_top_node_records(dapol_ctx* ctx, int32_t world, const uint8_t* records, size_t n, const uint8_t* level_above,
                                     const uint64_t* index, uint8_t* C32, uint8_t* H32, uint64_t* v, uint8_t* r32);
int32_t dapol_prove_batch_records(dapol_ctx* ctx, size_t k, const uint64_t* leaf_idx, size_t n_siblings, const uint8_t* sib_C32,
                                  const uint64_t* sib_v, const uint8_t* sib_r32, int32_t policy, int32_t aggregation_factor,
                                  int32_t n_bits, const uint8_t nonce_seed32[32], uint8_t* range_out);

/* Bench / roofline support: device-resident variant of build + prove-all used by bench.py so that the timed
 * region starts with inputs already in HBM and nothing is copied back.  Handles are opaque device buffers. */
typedef struct dapol_workload dapol_workload;
int32_t dapol_workload_create(dapol_ctx* ctx, int32_t height, size_t n, const uint64_t* leaf_idx, const uint64_t* v,
                              const uint8_t* r32, dapol_workload** out);
/* Sharded variant: this GPU holds the leaves of one top-level subtree (see dapol_tree_build_shard). */
int32_t dapol_workload_create_shard(dapol_ctx* ctx, int32_t total_height, int32_t shard_bits, size_t n, const uint64_t* leaf_idx,
                                    const uint64_t* v, const uint8_t* r32, dapol_workload** out);
int32_t dapol_workload_destroy(dapol_workload* w);
int32_t dapol_workload_tree(dapol_workload* w, dapol_tree** out);    /* borrowed: the tree of the last dapol_workload_build */
/* One pass: tree build + one padding-policy inclusion range proof per entity (aggregation_factor = height).
 * Returns device time of the two phases in milliseconds (HIP events on the ctx stream), the time and launch
 * count of the dominant kernel (the fixed-base MSM), and a 64-bit checksum of all proof bytes + the root. */
typedef struct {
    double tree_ms, prove_ms, msm_ms;
    uint64_t msm_launches, proofs, proof_bytes;
    uint64_t checksum;
    uint8_t root_C[32], root_H[32];
    /* (appended in round 3; the fields above keep their offsets)  msm_ms / msm_launches bracket the PLAIN fixed-base MSMs (S
     * commitment + never-fold rounds), mat_ms / mat_launches the materialisation of the folded generators; *_kernels = launches
     * of the dominant kernel inside those brackets (a generator-stationary MSM is a few hundred tile launches of k_rp_msm_gs). */
    double mat_ms;
    uint64_t mat_launches, msm_kernels, mat_kernels;
    double msm_all_ms;      /* time with a bracket of either kind open (msm_ms / mat_ms / this are UNIONS of the bracket intervals:
                             * two chunks are in flight on two streams and their brackets overlap) */
    double msm_span_ms;     /* SUM of the plain brackets' lengths: what a per-kernel profiler adds up for launches that overlap in time */
} dapol_workload_stats;
int32_t dapol_workload_run(dapol_workload* w, const uint8_t pad_seed32[32], const uint8_t nonce_seed32[32], int32_t n_bits,
                           size_t first_entity, size_t n_entities, dapol_workload_stats* stats);
/* The two halves of dapol_workload_run, so that the subtree-root exchange can sit between them: build returns the
 * (sub)tree root record; prove takes the siblings above it (n_upper may be 0).  Both fill their part of *stats. */
int32_t dapol_workload_build(dapol_workload* w, const uint8_t pad_seed32[32], uint8_t root_C[32], uint8_t root_H[32], uint64_t* root_v,
                             uint8_t root_r[32], dapol_workload_stats* stats);
int32_t dapol_workload_prove(dapol_workload* w, const uint8_t nonce_seed32[32], int32_t n_bits, size_t first_entity, size_t n_entities,
                             int32_t n_upper, const uint8_t* up_C32, const uint8_t* up_H32, const uint64_t* up_v, const uint8_t* up_r32,
                             dapol_workload_stats* stats);
/* dapol_workload_prove under either policy / any aggregation factor (the reference's bench also times RangeProofSplitting,
 * benches/dapol.rs:71-78); dapol_workload_prove = (DAPOL_POLICY_PADDING, height). */
int32_t dapol_workload_prove_policy(dapol_workload* w, const uint8_t nonce_seed32[32], int32_t n_bits, size_t first_entity, size_t n_entities,
                                    int32_t policy, int32_t aggregation_factor, int32_t n_upper, const uint8_t* up_C32, const uint8_t* up_H32,
                                    const uint64_t* up_v, const uint8_t* up_r32, dapol_workload_stats* stats);
/* Siblings of sampled leaves of the last build, with the upper siblings prepended: [b][n_upper+levels].  (v, r) are the
 * secrets handed to the range prover; (C, H) the Merkle path proof nodes.  Output / upper pointers may be NULL in pairs. */
int32_t dapol_workload_paths(dapol_workload* w, size_t b, const uint64_t* leaf_idx, int32_t n_upper, const uint64_t* up_v,
                             const uint8_t* up_r32, const uint8_t* up_C32, const uint8_t* up_H32, uint64_t* sib_v, uint8_t* sib_r32,
                             uint8_t* sib_C32, uint8_t* sib_H32);
/* Copies back the proofs of entities [first, first+count) of the last run (count*proof_size bytes). */
int32_t dapol_workload_proofs(dapol_workload* w, size_t first, size_t count, uint8_t* out);

#ifdef __cplusplus
}
#endif
#endif /* DAPOL_HIP_H */
