// Every byte string this library takes ON TRUST from the crates the reference depends on but does not vendor
// (bulletproofs 4.0.0, merlin 3.0.0; /root/reference/Cargo.toml:20-22) -- domain separators, transcript labels, the
// generator chain's seed.  They are restated from the published sources as remembered; nothing the reference repository
// holds pins them ("parity unpinned").  They live HERE and nowhere else in the product, so that a
// maintainer who diffs them against the crates fixes a mismatch in one line.  (The oracle under oracle/ keeps its own
// copies on purpose: it is an independent restatement.)
//
// LBL_X expands to TWO arguments: the literal and its length -- the (const char*, int) pair hash.h's Merlin helpers take.
#pragma once
#define DAPOL_LBL_(s) s, (int)(sizeof(s) - 1)

// merlin 3.0.0 -- src/transcript.rs, src/strobe.rs
#define LBL_STROBE_PROTO       DAPOL_LBL_("Merlin v1.0")        // Strobe128::new(MERLIN_PROTOCOL_LABEL)
#define LBL_DOM_SEP            DAPOL_LBL_("dom-sep")            // Transcript::new: append_message(b"dom-sep", label)
// the reference's application label: Transcript::new(&[])  (src/range/mod.rs:51,67,86,105)
#define LBL_APP_TRANSCRIPT     DAPOL_LBL_("")

// bulletproofs 4.0.0 -- src/transcript.rs (TranscriptProtocol)
#define LBL_RANGEPROOF_DOMAIN  DAPOL_LBL_("rangeproof v1")      // rangeproof_domain_sep: dom-sep, then "n", "m" as u64
#define LBL_IPP_DOMAIN         DAPOL_LBL_("ipp v1")             // innerproduct_domain_sep: dom-sep, then "n" as u64
#define LBL_N                  DAPOL_LBL_("n")
#define LBL_M                  DAPOL_LBL_("m")
// bulletproofs 4.0.0 -- src/range_proof/{dealer,mod}.rs, src/inner_product_proof.rs
#define LBL_V                  DAPOL_LBL_("V")                  // append_point per party's value commitment
#define LBL_A                  DAPOL_LBL_("A")
#define LBL_S                  DAPOL_LBL_("S")
#define LBL_Y                  DAPOL_LBL_("y")                  // challenge_scalar
#define LBL_Z                  DAPOL_LBL_("z")
#define LBL_T1                 DAPOL_LBL_("T_1")
#define LBL_T2                 DAPOL_LBL_("T_2")
#define LBL_X                  DAPOL_LBL_("x")
#define LBL_TX                 DAPOL_LBL_("t_x")                // append_scalar
#define LBL_TX_BLINDING        DAPOL_LBL_("t_x_blinding")
#define LBL_E_BLINDING         DAPOL_LBL_("e_blinding")
#define LBL_W                  DAPOL_LBL_("w")
#define LBL_L                  DAPOL_LBL_("L")
#define LBL_R                  DAPOL_LBL_("R")
#define LBL_U                  DAPOL_LBL_("u")
// bulletproofs 4.0.0 -- src/generators.rs: SHAKE256(GENERATORS_CHAIN_SEED || 'G' / 'H' || u32le(party))
#define LBL_GENERATORS_CHAIN   DAPOL_LBL_("GeneratorsChain")
#define LBL_GENS_G             'G'
#define LBL_GENS_H             'H'

// This library's own (not from any crate): the squeeze that closes a verifier transcript for the batch weights, after the
// final a, b have been appended (kernels_verify.h, k_rv_transcript).
#define LBL_OWN_A              DAPOL_LBL_("a")
#define LBL_OWN_B              DAPOL_LBL_("b")
#define LBL_OWN_BATCH          DAPOL_LBL_("dapol-batch")
