/* vmnproofs.h — proof-level C ABI (seam S1 of SURVEY.md §8b): the sigma-protocol cores and the shuffler's
 * arithmetic lines of the reference as host-side C++ drivers (verificatum-vmn_amd/csrc/vmnproofs.cpp) that
 * issue array operations through vmnhip.h.  A Java `PoSGPU / PoSCGPU / CCPoSGPU` triple binds
 * these entry points one to one; nothing here touches the GPU except through vmnhip.h.
 *
 * Classes mirrored (same operation order, same message item order, same verdict rules):
 *   vmn_pos_*    PoSBasicTW    src/java/com/verificatum/protocol/hvzk/PoSBasicTW.java
 *                (precompute :436-482 / :394-402, setBatchVector :533-538, commit :546-700, reply :856-888,
 *                 computeAF :407-410, setCommitment :780-823, setChallenge :840-847, verify :1000-1066)
 *   vmn_posc_*   PoSCBasicTW   hvzk/PoSCBasicTW.java (setInstance :306-340, commit :363-529, reply :607-636,
 *                 verify :646-727, short-circuiting)
 *   vmn_ccpos_*  CCPoSBasicW   hvzk/CCPoSBasicW.java (commit :344-396, reply :462-485, computeAB :493-506,
 *                 verify :519-584, plain and "raised" form)
 *   vmn_shuffle_reencrypt        mixnet/ShufflerElGamalSession.java:400-409, 273-278 (_reencryption_factors / _apply_factors: :645-661, :789-792)
 *   vmn_permutation_commitment   mixnet/PermutationCommitment.java:189-215
 *   vmn_decryption_factors, vmn_combine_decryption_factors, vmn_decproof_*
 *                                elgamal/DistrElGamalSession.java:365-385, elgamal/DistrElGamalSessionBasic.java
 *   vmn_igen_*                   distr/IndependentGeneratorsBasicI.java (interactive independent generators)
 *   vmn_element_*                single group elements on the host (what VCR keeps in PGroupElement objects)
 *
 * Conventions
 *   * A ciphertext array of width w is 2w component arrays [u_1..u_w, v_1..v_w] (struct of arrays, the way
 *     PPGroupElementArray.project exposes it, elgamal/DistrElGamalSession.java:377-378); the wide public key is
 *     the matching 2w group elements [g..g, y..y] (elgamal/ProtocolElGamal.java:785-800), elem_bytes each.
 *   * Arrays handed IN are borrowed (the caller keeps them alive until the proof object is freed); arrays handed
 *     OUT inside a vmn_msg belong to the message.
 *   * Single group elements / ring elements cross as big-endian bytes of vmn_group_elem_bytes / exp_bytes.
 *   * Status codes are those of vmnhip.h; a failed verification is verdict 0 with status VMN_OK, never an error
 *     (PoSBasicTW.java:1065).  vmn_last_error() of vmnhip.h holds the message of the last failure.
 *   * VCR's PRG / randomElementArray sampling is not part of the reference tree: randomness
 *     and the batching vector are explicit inputs (vmn_random_source, *_set_batch_vector).
 */
#ifndef VMNPROOFS_H
#define VMNPROOFS_H

#include "vmnhip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The prover's randomness, at the granularity the reference draws it: PRing.randomElementArray / randomElement
 * (PoSBasicTW.java:446, 465, 583, 612, 667, 673, 687) and LargeIntegerArray.random (:470-475).
 *
 * Host rows: each of the first two callbacks sets *rows to n big-endian rows of vmn_group_exp_bytes() bytes, owned by the
 * source and valid until its next call; integers of `bits` bits are delivered as field elements (reduced mod q when bits
 * exceeds the order).  Return 0 on success.
 *
 * Device arrays: when `array_seed` is not NULL, every N-sized draw (r, s, b, beta, epsilon: N > 1) asks the source for
 * 32 fresh bytes only and expands them ON THE DEVICE (PRGHeuristic over SHA-256, csrc k_prg_rows, the generator
 * vmn_rarray_from_prg uses): ring elements = the i-th ceil((bits(q) + rbitlen) / 8) bytes of PRG(seed), leading bits
 * cleared, reduced mod q (statistical distance 2^-rbitlen from uniform, the meaning of the reference's rbitlen,
 * ProtocolElGamal.java:277-306); integers of `bits` bits = the i-th ceil(bits / 8) bytes, leading bits cleared.  No N-sized
 * host buffer, no upload.  The 32 bytes must come from the party's cryptographic RandomSource (the reference's
 * `randomSource`, e.g. /dev/urandom behind RandomDevice, demo/mixnet/.checkbaseconf:124); VCR's own procedure inside
 * randomElementArray is not in the reference tree, and a prover's private randomness needs no interoperability.  Single
 * elements (alpha, gamma, ...) always come through `ring_elements`. */
typedef struct vmn_random_source {
    void* user;
    int (*ring_elements)(void* user, size_t n, const uint8_t** rows);
    int (*integers)(void* user, size_t n, int bits, const uint8_t** rows);
    int (*array_seed)(void* user, uint8_t seed_out[32]);          /* optional (NULL = host rows for arrays too) */
} vmn_random_source;
/* pRing.randomElementArray(n, randomSource, rbitlen) for a caller that needs such an array itself -- the re-encryption
 * exponents, ShufflerElGamalSession.java:400-409 (:408-409); the commitment exponents, PermutationCommitment.java:189-199. */
int vmn_rarray_random(vmn_group* grp, const vmn_random_source* rs, size_t n, int rbitlen, vmn_rarray** out);

/* ---- messages: ordered items, each an element array, a ring array, k group elements or k ring elements ----
 *   PoS  commitment (PoSBasicTW.java:694-699):  B[N], A', B'[N], C', D', F'[2w]
 *   PoS  reply      (:880-886):                 k_A, k_B[N], k_C, k_D, k_E[N], k_F[w]
 *   PoSC commitment (PoSCBasicTW.java:524-528): B[N], A', B'[N], C', D'
 *   PoSC reply      (:629-634):                 k_A, k_B[N], k_C, k_D, k_E[N]
 *   CCPoS commitment (CCPoSBasicW.java:395):    A', B'[2w]
 *   CCPoS reply      (:480-483):                k_A, k_B[w], k_E[N]                                           */
typedef struct vmn_msg vmn_msg;
enum { VMN_ITEM_GARRAY = 1, VMN_ITEM_RARRAY = 2, VMN_ITEM_ELEMENTS = 3, VMN_ITEM_RING = 4 };
int vmn_msg_create(vmn_msg** out);
/* the same for a message whose group elements belong to grp: over a curve group the single elements of the wire form are
 * node(leaf(x), leaf(y)) instead of a leaf (messages made by the drivers and by vmn_msg_from_bytetree know their group) */
int vmn_msg_create_for(vmn_group* grp, vmn_msg** out);
void vmn_msg_free(vmn_msg* m);                                   /* frees the arrays it owns */
size_t vmn_msg_items(const vmn_msg* m);
int vmn_msg_item_kind(const vmn_msg* m, size_t i);
const vmn_garray* vmn_msg_item_garray(const vmn_msg* m, size_t i);
const vmn_rarray* vmn_msg_item_rarray(const vmn_msg* m, size_t i);
int vmn_msg_item_bytes(const vmn_msg* m, size_t i, const uint8_t** data, size_t* count, size_t* width);
/* receiving side: ownership of the array passes to the message */
int vmn_msg_push_garray(vmn_msg* m, vmn_garray* a);
int vmn_msg_push_rarray(vmn_msg* m, vmn_rarray* a);
int vmn_msg_push_elements(vmn_msg* m, const uint8_t* be, size_t count, size_t width);
int vmn_msg_push_ring(vmn_msg* m, const uint8_t* be, size_t count, size_t width);
/* Wire form: node(items); an array item is the array's byte tree, one element a leaf, k > 1
 * elements of a ciphertext-shaped item node(node(k/2 leaves), node(k/2 leaves)) (k = 2: node(leaf, leaf)), k > 1
 * ring elements node(k leaves).  Over ECqPGroup an element is node(leaf(x), leaf(y)) (infinity: both -1) and an array
 * node(N such nodes): [NOT-IN-REF] -- VCR's ECqPGroupElement / BPGroupElementArray, restated from the verifier
 * specification, not pinned by a reference fixture. */
size_t vmn_msg_bytetree_size(const vmn_msg* m);
int vmn_msg_to_bytetree(const vmn_msg* m, uint8_t* out);
/* `layout` lists the expected item kinds (VMN_ITEM_*), `counts[i]` the element count of items of kind 3 / 4 and
 * the array size of kinds 1 / 2.  *format_ok = 0 (and no message) on malformed input -- framing, a value out of
 * range, an array element outside the subgroup (vmn_garray_is_member: the check pGroup.toElementArray makes): the
 * caller substitutes trivial values as the reference does (PoSBasicTW.java:794-815). */
int vmn_msg_from_bytetree(vmn_group* grp, const uint8_t* bt, size_t len, const int* layout, const size_t* counts,
                          size_t items, vmn_msg** out, int* format_ok);

/* ---- one proof over several GPUs (BASELINE.json configs[3], [4]) ---------------------------------------
 * One process per GPU.  Every rank creates the same proof object, sets the same communicator (its own rank), and makes
 * the SAME calls in the same order with random sources that return the SAME values on every rank (one party owns all
 * ranks: it seeds them alike).  Position-indexed arrays are split into contiguous shards (vmn_shard_bounds); each rank
 * computes and keeps its shard of u, w', B, B', k_B, k_E.  Arguments:
 *   h                       always the WHOLE array, replicated on every rank (provers read it through the permutation;
 *                           its size is the size of the proof);
 *   u, w, w', r, s, raised  the whole array or this rank's shard -- told apart by their size;
 *   pi                      the whole permutation table;
 *   messages                scalars are identical on every rank, array items are this rank's shard.
 * h and w are read through the permutation as local gathers, so no element crosses a link.  The batching vector (from its
 * seed) and the prover's N-sized random arrays (vmn_random_source.array_seed) are counter-mode PRG streams: a rank
 * generates only its own positions and the rows it reads through the permutation (vmn_rarray_from_prg_range / _gather),
 * O(N / world) work per rank; handed over as host rows they are kept whole.  The exchanges are fixed-size all-gathers of scalars:
 * partial products of expProd / prod, partial sums, scan carries, each shard's last B, verdict bits -- at most one per
 * phase and a few hundred bytes per rank ("all-reduce" of north_star: modular multiplication is not a reduction operator
 * of RCCL, so it is all-gather + local multiplication).
 * all_gather: every rank contributes `bytes` bytes; recv receives world * bytes in rank order; 0 on success. */
typedef struct vmn_comm {
    void* user;
    int rank, world;
    int (*all_gather)(void* user, const uint8_t* send, size_t bytes, uint8_t* recv);
} vmn_comm;
/* shard k of n positions over `world` ranks: [lo, hi), sizes differ by at most one (possibly empty) */
void vmn_shard_bounds(size_t n, int world, int rank, size_t* lo, size_t* hi);
/* this rank's positions of the shuffler's two arrays, gathered out of the whole inputs (no exchange):
 * w'[lo, hi) of permute(w * pk^s, pi^-1) and u[lo, hi) of permute(h * g^r, pi). */
int vmn_shuffle_reencrypt_shard(vmn_group* grp, const uint8_t* pkey_be, size_t width, const vmn_garray* const* w_full,
                                const vmn_rarray* const* s_full, const uint32_t* pi, size_t lo, size_t hi, vmn_garray** wp_out);
int vmn_permutation_commitment_shard(vmn_group* grp, const uint8_t* g_be, const vmn_garray* h_full, const vmn_rarray* r_full,
                                     const uint32_t* pi, size_t lo, size_t hi, vmn_garray** u_out);
/* The same for exponents that are PRG draws: rs->array_seed is asked for one seed per column (the order vmn_rarray_random
 * would ask in), and the rank generates just the rows it reads -- s_{pi^-1(i)} resp. r_{pi(i)} for i in [lo, hi) -- and its
 * own positions s[lo, hi) resp. r[lo, hi), returned for the proof object (s_out: `width` arrays).  Without array_seed the
 * source's host rows are used (whole arrays).  ref: ShufflerElGamalSession.java:400-409, PermutationCommitment.java:189-215. */
int vmn_shuffle_reencrypt_shard_seeded(vmn_group* grp, const uint8_t* pkey_be, size_t width, const vmn_garray* const* w_full,
                                       const vmn_random_source* rs, int rbitlen, const uint32_t* pi, size_t lo, size_t hi,
                                       vmn_garray** wp_out, vmn_rarray** s_out);
int vmn_permutation_commitment_shard_seeded(vmn_group* grp, const uint8_t* g_be, const vmn_garray* h_full, const vmn_random_source* rs,
                                            int rbitlen, const uint32_t* pi, size_t lo, size_t hi, vmn_garray** u_out, vmn_rarray** r_out);

/* ---- what a binding needs to size its host buffers (jni/: the generated wrappers check every array against the bytes the
 * callee reads or writes): the group of a proof object, the number of rows a batching vector handed to it must have (all N
 * positions of the proof; 0 before the instance is known), the number of parties of the threshold objects. */
typedef struct vmn_pos vmn_pos;
typedef struct vmn_posc vmn_posc;
typedef struct vmn_ccpos vmn_ccpos;
typedef struct vmn_decproof vmn_decproof;
typedef struct vmn_igen vmn_igen;
vmn_group* vmn_pos_group(const vmn_pos* p);
vmn_group* vmn_posc_group(const vmn_posc* p);
vmn_group* vmn_ccpos_group(const vmn_ccpos* p);
vmn_group* vmn_decproof_group(const vmn_decproof* p);
vmn_group* vmn_igen_group(const vmn_igen* p);
size_t vmn_pos_size(const vmn_pos* p);
size_t vmn_posc_size(const vmn_posc* p);
size_t vmn_ccpos_size(const vmn_ccpos* p);
size_t vmn_decproof_size(const vmn_decproof* p);
size_t vmn_igen_size(const vmn_igen* p);
int vmn_decproof_parties(const vmn_decproof* p);      /* k */
int vmn_igen_parties(const vmn_igen* p);              /* threshold */

/* ---- PoSBasicTW --------------------------------------------------------------------------------------------- */
typedef struct vmn_pos vmn_pos;
/* rs = NULL for a verifier.  ref: constructor PoSBasicTW.java:300-330 (vbitlen, ebitlen, rbitlen). */
int vmn_pos_create(vmn_group* grp, int vbitlen, int ebitlen, int rbitlen, const vmn_random_source* rs, vmn_pos** out);
void vmn_pos_free(vmn_pos* p);                                    /* free() :1088-1101 */
/* sharded proof: before precompute (see vmn_comm above) */
int vmn_pos_set_comm(vmn_pos* p, const vmn_comm* comm);
/* prover (pi != NULL, :436-482): draws r, computes u = permute(h g^r, pi), alpha, epsilon, A'.
 * verifier (pi == NULL, :394-402): records g, h. */
int vmn_pos_precompute(vmn_pos* p, const uint8_t* g_be, const vmn_garray* h, const uint32_t* pi);
const vmn_garray* vmn_pos_permutation_commitment(const vmn_pos* p);          /* u, owned by the proof object */
int vmn_pos_set_permutation_commitment(vmn_pos* p, const vmn_garray* u);     /* verifier :780-792 */
/* setInstance :421-433; s = the w arrays of re-encryption exponents (prover) or NULL (verifier) */
int vmn_pos_set_instance(vmn_pos* p, const uint8_t* pkey_be, size_t width, const vmn_garray* const* w,
                         const vmn_garray* const* wp, const vmn_rarray* const* s);
int vmn_pos_set_batch_vector(vmn_pos* p, const uint8_t* e_be);    /* N rows of exp_bytes; :533-538 */
/* setBatchVector(byte[] prgSeed) as the reference has it (:533-538): e is derived on the GPU from the 32-byte seed
 * (vmn_rarray_from_prg: PRGHeuristic over SHA-256, N integers of ebitlen bits).  Same for PoSC / CCPoS below. */
int vmn_pos_set_batch_vector_seed(vmn_pos* p, const uint8_t* seed, size_t seedlen);
/* Optional: the part of commit() that does not depend on the batching vector -- its random draws (same order and values)
 * and the multi-exponentiations with epsilon.  The reference derives the batching vector by hashing the whole instance
 * (hvzk/PoSTW.java:118-130, hvzk/ChallengerRO.java:96-116: 1.5 GB of byte trees at N = 10^6, ~0.7 s of SHA-256 on one
 * core); a caller runs this part on the GPU while its hash thread -- the helper thread of vmn_ctx_helper_begin, which
 * exports the byte trees on its own stream -- computes the seed.  commit() runs it itself when it was not called. */
int vmn_pos_commit_prepare(vmn_pos* p);
int vmn_pos_commit(vmn_pos* p, vmn_msg** commitment);             /* :546-700 */
int vmn_pos_reply(vmn_pos* p, const uint8_t* v_be, size_t vbytes, vmn_msg** reply);   /* :856-888 */
int vmn_pos_compute_af(vmn_pos* p);                               /* :407-410 */
int vmn_pos_set_commitment(vmn_pos* p, const vmn_msg* commitment);/* :780-823 (parsed form) */
int vmn_pos_set_challenge(vmn_pos* p, const uint8_t* v_be, size_t vbytes);            /* :840-847 */
/* The part of verify() that needs the reply but not the challenge -- the right side of check (B), the multi-exponentiations
 * with k_E, g^{k_A}, g^{k_C}, g^{k_D}, pk^{-k_F}: four fifths of the verifier's work -- so that it can run while the
 * challenge is still being derived (the verifier hashes the whole commitment for it; the reply is published long
 * before).  Needs computeAF and setCommitment.  Optional: verify() does this part itself when it was not called for the
 * same reply. */
int vmn_pos_verify_prepare(vmn_pos* p, const vmn_msg* reply);
/* :1000-1066; all five checks are evaluated; verdicts5 (may be NULL) = A, B, C, D, F */
int vmn_pos_verify(vmn_pos* p, const vmn_msg* reply, int* verdict, int* verdicts5);
/* The verifier's intermediate values, as the reference exposes them: getA PoSBasicTW.java:716, getF :761 (after computeAF;
 * 2 * width elements), getC :949, getD :958 (after verify).  `vmnv -t PoS.A,PoS.F,PoS.C,PoS.D` prints exactly these
 * (mixnet/MixNetElGamalVerifyFiatShamirSession.java:880-932): tools/vmnv_vectors.py does the same from a proof directory.
 * The other getters of the reference (getB, getAp ... getFp, getk_A ... getk_F, :707-770, 895-940) are the items of the
 * commitment / reply messages (vmn_msg_scalar, vmn_msg_garray, vmn_msg_rarray).  out_be: elem_bytes per element. */
int vmn_pos_get_A(vmn_pos* p, uint8_t* out_be);
int vmn_pos_get_F(vmn_pos* p, uint8_t* out_be);
int vmn_pos_get_C(vmn_pos* p, uint8_t* out_be);
int vmn_pos_get_D(vmn_pos* p, uint8_t* out_be);
size_t vmn_pos_width(const vmn_pos* p);              /* omega of the instance (0 before set_instance) */

/* ---- PoSCBasicTW -------------------------------------------------------------------------------------------- */
typedef struct vmn_posc vmn_posc;
int vmn_posc_create(vmn_group* grp, int vbitlen, int ebitlen, int rbitlen, const vmn_random_source* rs, vmn_posc** out);
void vmn_posc_free(vmn_posc* p);
int vmn_posc_set_comm(vmn_posc* p, const vmn_comm* comm);          /* before set_instance */
/* setInstance :306-340; r, pi = NULL for a verifier */
int vmn_posc_set_instance(vmn_posc* p, const uint8_t* g_be, const vmn_garray* h, const vmn_garray* u,
                          const vmn_rarray* r, const uint32_t* pi);
int vmn_posc_set_batch_vector(vmn_posc* p, const uint8_t* e_be);
int vmn_posc_set_batch_vector_seed(vmn_posc* p, const uint8_t* seed, size_t seedlen);    /* :350-355 */
int vmn_posc_commit_prepare(vmn_posc* p);                         /* see vmn_pos_commit_prepare */
int vmn_posc_commit(vmn_posc* p, vmn_msg** commitment);           /* :363-529 */
int vmn_posc_reply(vmn_posc* p, const uint8_t* v_be, size_t vbytes, vmn_msg** reply);  /* :607-636 */
int vmn_posc_set_commitment(vmn_posc* p, const vmn_msg* commitment);
int vmn_posc_set_challenge(vmn_posc* p, const uint8_t* v_be, size_t vbytes);
int vmn_posc_verify_prepare(vmn_posc* p, const vmn_msg* reply);                      /* the reply side of verify(), see vmn_pos_verify_prepare */
int vmn_posc_verify(vmn_posc* p, const vmn_msg* reply, int* verdict);                 /* :646-727 */
/* A = u.expProd(e) (:676), C (:718-723), D (:724-727) of the verifier, after verify / verify_prepare (private fields in the
 * reference; exposed for the intermediate-value tests and the test-vector dump) */
int vmn_posc_get_A(vmn_posc* p, uint8_t* out_be);
int vmn_posc_get_C(vmn_posc* p, uint8_t* out_be);
int vmn_posc_get_D(vmn_posc* p, uint8_t* out_be);

/* ---- CCPoSBasicW -------------------------------------------------------------------------------------------- */
typedef struct vmn_ccpos vmn_ccpos;
int vmn_ccpos_create(vmn_group* grp, int vbitlen, int ebitlen, int rbitlen, const vmn_random_source* rs, vmn_ccpos** out);
void vmn_ccpos_free(vmn_ccpos* p);
int vmn_ccpos_set_comm(vmn_ccpos* p, const vmn_comm* comm);        /* before set_instance */
/* setInstance :290-330; r, pi, s = NULL for a verifier */
int vmn_ccpos_set_instance(vmn_ccpos* p, const uint8_t* g_be, const vmn_garray* h, const vmn_garray* u,
                           const uint8_t* pkey_be, size_t width, const vmn_garray* const* w,
                           const vmn_garray* const* wp, const vmn_rarray* r, const uint32_t* pi,
                           const vmn_rarray* const* s);
int vmn_ccpos_set_batch_vector(vmn_ccpos* p, const uint8_t* e_be);
int vmn_ccpos_set_batch_vector_seed(vmn_ccpos* p, const uint8_t* seed, size_t seedlen);  /* :330-335 */
int vmn_ccpos_commit_prepare(vmn_ccpos* p);                       /* see vmn_pos_commit_prepare */
int vmn_ccpos_commit(vmn_ccpos* p, vmn_msg** commitment);         /* :344-396 */
int vmn_ccpos_reply(vmn_ccpos* p, const uint8_t* v_be, size_t vbytes, vmn_msg** reply);  /* :462-485 */
int vmn_ccpos_set_commitment(vmn_ccpos* p, const vmn_msg* commitment);
int vmn_ccpos_set_challenge(vmn_ccpos* p, const uint8_t* v_be, size_t vbytes);
/* computeAB :493-506; raisedu = u^rho selects the single-equation form (NULL = plain) */
int vmn_ccpos_compute_ab(vmn_ccpos* p, const vmn_garray* raisedu);
/* The values of computeAB: plain form A then B (1 + 2 width elements), raised form AB (2 width); *count = elements written
 * (out_be must hold 1 + 2 width).  Private fields in the reference (CCPoSBasicW.java:493-506). */
int vmn_ccpos_get_AB(vmn_ccpos* p, uint8_t* out_be, size_t* count);
size_t vmn_ccpos_width(const vmn_ccpos* p);
/* verify :519-584; raisedh / rho_be = NULL for the plain form */
/* The reply side of verify() (here: ALL its array work, the multi-exponentiations with k_E); same raisedh / rho as the
 * verify() that follows.  See vmn_pos_verify_prepare. */
int vmn_ccpos_verify_prepare(vmn_ccpos* p, const vmn_msg* reply, const vmn_garray* raisedh, const uint8_t* rho_be, size_t rho_bytes);
int vmn_ccpos_verify(vmn_ccpos* p, const vmn_msg* reply, const vmn_garray* raisedh, const uint8_t* rho_be,
                     size_t rho_bytes, int* verdict);

/* ---- verifiable threshold decryption ------------------------------------------------------
 * elgamal/DistrElGamalSession.java:365-385 (decryption factors f_j = u^(-x_j / c)), :536-538 (plaintexts);
 * elgamal/DistrElGamalSessionBasic.java: prodFactor :318-344, modifiedLagrangeCoefficients :358-452,
 * combineDecryptionFactors :465-503, setBatchVector :513-518, batchInput :524-526, commit :534-540, reply :595-598,
 * combine :642-678, batchCombined :683-685, verifyCombined :693-700, batch :707-709, verify :718-727.
 * Parties are numbered 1..k; arrays of per-party values have k + 1 entries with entry 0 unused. */
int vmn_prod_factor(vmn_group* grp, int k, uint8_t* c_be);                                   /* c mod q, exp_bytes */
/* threshold coefficients of smallest absolute value for the first `threshold` correct parties: |lambda_t| as
 * exp_bytes rows and a sign flag each (1 = negative).  VMN_ERR_ARG when fewer than `threshold` parties are correct. */
int vmn_lagrange_coefficients(vmn_group* grp, const uint8_t* correct, int k, int threshold, uint8_t* abs_be, int* negative);
int vmn_decryption_factors(vmn_group* grp, const vmn_garray* u, const uint8_t* secret_be, int k, vmn_garray** f_out);
/* out[i] = prod_t f_{j_t}[i]^(lambda_t): negative coefficients go through one batch inversion */
int vmn_combine_decryption_factors(vmn_group* grp, const vmn_garray* const* f, const uint8_t* correct, int k, int threshold,
                                   vmn_garray** out);
/* plaintexts = v.mul(combinedFactors) is vmn_garray_mul. */

typedef struct vmn_decproof vmn_decproof;          /* DistrElGamalSessionBasic of party j (prover and verifier of all l) */
int vmn_decproof_create(vmn_group* grp, int j, int k, int threshold, int ebitlen, const vmn_random_source* rs, vmn_decproof** out);
void vmn_decproof_free(vmn_decproof* p);
/* u: first components; y_be: k + 1 public key shares g^(x_l) (entry 0 unused); f: k + 1 factor arrays (NULL = absent) */
int vmn_decproof_set_instance(vmn_decproof* p, const vmn_garray* u, const uint8_t* y_be, const vmn_garray* const* f);
int vmn_decproof_set_batch_vector(vmn_decproof* p, const uint8_t* e_be);
int vmn_decproof_set_batch_vector_seed(vmn_decproof* p, const uint8_t* seed, size_t seedlen);
int vmn_decproof_batch_input(vmn_decproof* p);                                               /* A = u.expProd(e) */
int vmn_decproof_commit(vmn_decproof* p, const uint8_t* x_be, uint8_t* yp_out, uint8_t* Bp_out);   /* :534-540 */
int vmn_decproof_reply(vmn_decproof* p, const uint8_t* v_be, size_t vbytes, uint8_t* kx_out);      /* :595-598 */
int vmn_decproof_set_commitment(vmn_decproof* p, int l, const uint8_t* yp_be, const uint8_t* Bp_be);
int vmn_decproof_set_reply(vmn_decproof* p, int l, const uint8_t* kx_be);
int vmn_decproof_batch(vmn_decproof* p, int l);                                              /* B_l = f_l.expProd(e) */
int vmn_decproof_verify(vmn_decproof* p, int l, const uint8_t* v_be, size_t vbytes, int* verdict);           /* :718-727 */
int vmn_decproof_combine(vmn_decproof* p, const uint8_t* correct, const uint8_t* combinedy_be, const vmn_garray* combinedf);
int vmn_decproof_batch_combined(vmn_decproof* p);                                            /* :683-685 */
int vmn_decproof_verify_combined(vmn_decproof* p, const uint8_t* v_be, size_t vbytes, int* verdict);         /* :693-700 */

/* ---- interactive derivation of independent generators -----------------------------------
 * distr/IndependentGeneratorsBasicI.java: setInstance :166-175, setBatchVector :186-193, commit :201-208,
 * setCommitment :219-227, setChallenge :235-238, reply :245-248, setReply :259-267, verify() :275-289 (combined),
 * verify(l) :297-299.  Party j proves knowledge of the exponents s of its generator parts h_j = g^s:
 * a = <s, e>, A'_j = g^r, k_a = a v + r;  h_l.expProd(e)^v A'_l = g^(k_a,l).  Parties 1..threshold; arrays of
 * per-party values have threshold + 1 entries, entry 0 unused. */
typedef struct vmn_igen vmn_igen;
int vmn_igen_create(vmn_group* grp, int j, int threshold, int ebitlen, const vmn_random_source* rs, vmn_igen** out);
void vmn_igen_free(vmn_igen* p);
/* h: threshold + 1 arrays of generator parts (NULL = absent); s: this party's exponents (NULL for a pure verifier);
 * combinedh: the product of the parts */
int vmn_igen_set_instance(vmn_igen* p, const uint8_t* g_be, const vmn_garray* const* h, const vmn_rarray* s, const vmn_garray* combinedh);
int vmn_igen_set_batch_vector(vmn_igen* p, const uint8_t* e_be);
int vmn_igen_set_batch_vector_seed(vmn_igen* p, const uint8_t* seed, size_t seedlen);
int vmn_igen_commit(vmn_igen* p, uint8_t* Ap_out);
int vmn_igen_set_commitment(vmn_igen* p, int l, const uint8_t* Ap_be);     /* not a group element: the unit is taken (:223-226) */
int vmn_igen_set_challenge(vmn_igen* p, const uint8_t* v_be, size_t vbytes);
int vmn_igen_reply(vmn_igen* p, uint8_t* ka_out);
int vmn_igen_set_reply(vmn_igen* p, int l, const uint8_t* ka_be);          /* out of range: zero is taken (:263-266) */
int vmn_igen_verify_combined(vmn_igen* p, int* verdict);
int vmn_igen_verify(vmn_igen* p, int l, int* verdict);

/* ---- single group elements on the host (what the drivers above use for A', C', ... ; exposed for callers that hold
 * such elements themselves, e.g. the sharded multi-GPU proof driver): big-endian elem_bytes in and out; the exponent
 * is a non-negative big-endian integer of any length.  ModPGroup: 64-bit Montgomery arithmetic; curves: Jacobian
 * arithmetic (csrc/hostnum64.h, csrc/hostcurve.h).  No GPU work. */
int vmn_element_exp(vmn_group* grp, const uint8_t* base_be, const uint8_t* e_be, size_t ebytes, uint8_t* out_be);
int vmn_element_mul(vmn_group* grp, const uint8_t* a_be, const uint8_t* b_be, uint8_t* out_be);
int vmn_element_inv(vmn_group* grp, const uint8_t* a_be, uint8_t* out_be);

/* ---- shuffler lines ----------------------------------------------------------------------------------------- */
/* w' = permute(w * pk^s, pi^-1): ShufflerElGamalSession.java:400-409 (widePublicKey.exp(reencExponents)), :273-278
 * (input.mul(reencFactors), permute(inverse), reencFactors.free()).  wp_out receives 2w new arrays. */
int vmn_shuffle_reencrypt(vmn_group* grp, const uint8_t* pkey_be, size_t width, const vmn_garray* const* w,
                          const vmn_rarray* const* s, const uint32_t* pi, vmn_garray** wp_out);
/* The same in the two steps of the reference's precomputed shuffle (`vmn -precomp`, then the committed shuffle): the
 * re-encryption factors pk^s -- ShufflerElGamalSession.java:645-661 (reencFactors = widePublicKey.exp(reencExponents),
 * written to file) -- and, when the ciphertexts arrive, w' = permute(w * factors, pi^-1) -- :789-792.  factors_out and
 * wp_out receive 2w new arrays each; fewer ciphertexts than were precomputed for: cut the factors with
 * vmn_garray_copy_range first (:673-712). */
int vmn_shuffle_reencryption_factors(vmn_group* grp, const uint8_t* pkey_be, size_t width, const vmn_rarray* const* s,
                                     vmn_garray** factors_out);
int vmn_shuffle_apply_factors(vmn_group* grp, size_t width, const vmn_garray* const* w, const vmn_garray* const* factors,
                              const uint32_t* pi, vmn_garray** wp_out);
/* u = permute(h * g^r, pi): PermutationCommitment.java:189-215 (:200 g.exp(exponents), :201 generators.mul, :215). */
int vmn_permutation_commitment(vmn_group* grp, const uint8_t* g_be, const vmn_garray* h, const vmn_rarray* r,
                               const uint32_t* pi, vmn_garray** u_out);

/* PermutationCommitment.shrink (mixnet/PermutationCommitment.java:390-471), the prover's side: the commitment was
 * precomputed for n_max ciphertexts (`vmn -precomp`, ShufflerElGamalSession.java:645-661) and only n <= n_max arrive
 * (:673-712: generators, raised generators, re-encryption exponents and factors are cut with copyOfRange(0, n) =
 * vmn_*_copy_range; the commitments with extract(keepList) = vmn_garray_extract).  keep_out[i] = 1 for the n positions
 * of u that commit to the first n generators (:398-405), pi_out = the permutation of [0, n) those positions carry
 * (Permutation.shrink).  pi is the table vmn_permutation_commitment was given (u[i] = (h g^r)[pi[i]]), so
 * keep_out[i] = (pi[i] < n).  Host work only.  The reference's keepList[permutation.map(i)] = true, i < n, is the same
 * list when pi is the table of its permutation's inverse -- which also settles App. B's open question: the positions
 * kept must be those of the first n generators (exponents and generators are cut to [0, n)), so VCR's
 * X.permute(pi) puts X[i] at position pi.map(i), and a JNI binding hands this library the inverse table. */
int vmn_permutation_shrink(const uint32_t* pi, size_t n_max, size_t n, uint8_t* keep_out, uint32_t* pi_out);
/* The verifiers' side (:424-447): a keep list read from another party is accepted only if it has n_max flags of which
 * exactly n are set; otherwise the trivial list (first n set) takes its place.  keep is rewritten in place; *replaced
 * (may be NULL) tells whether that happened.  keep_len = number of flags actually received. */
int vmn_keep_list_sanitize(uint8_t* keep, size_t keep_len, size_t n_max, size_t n, int* replaced);

#ifdef __cplusplus
}
#endif
#endif /* VMNPROOFS_H */
