/*
 * vmnhip.h — C ABI of the MI355X-native exponentiation / re-encryption / proof-of-shuffle core
 * for the Verificatum Mix-Net.
 *
 * This is the drop-in boundary.  The reference (verificatum-vmn, Java) reaches its
 * arithmetic through the array classes of VCR 3.1.0 (com.verificatum.arithm.PGroupElementArray,
 * PRingElementArray, LargeIntegerArray, ...), which are not part of the reference tree; the reference's
 * own files only show the *call sites*.  Every entry point below is one such array-level call, and
 * its comment cites the reference call site(s) it serves ("ref:" = path under /root/reference,
 * P/ = src/java/com/verificatum/protocol/).  A JNI shim maps `long` handles and `byte[]` onto these
 * functions one-to-one.
 *
 * Conventions
 *   - Plain C, no C++/torch types.  Handles are opaque pointers.
 *   - Every function returns VMN_OK (0) or a negative vmn_status; nothing throws or aborts
 *     (the reference replaces malformed inputs by trivial values and carries on:
 *     ref: P/hvzk/PoSBasicTW.java:794-815, P/mixnet/ShufflerElGamalSession.java:207-214).
 *   - Verdicts and membership results are returned through int* out-parameters (1 / 0).
 *   - Host byte buffers use the reference's wire format for fixed-width integers: big-endian,
 *     `nbytes` bytes per value, values concatenated (the payload of the byte-tree leaves,
 *     SURVEY.md App. D).  Arrays live on the device; results are fresh arrays (the reference's
 *     arrays are immutable, ref: P/hvzk/PoSBasicTW.java:1088-1101 for the explicit free()s).
 *   - All work is enqueued on the context's HIP stream; functions that return host data
 *     synchronise that stream.  Calls on distinct contexts are independent; calls on one context
 *     must be serialised by the caller (one protocol thread per party, SURVEY.md §8b "Threading").
 *   - There is NO CPU fallback: if no gfx950 device is usable, vmn_ctx_create fails with
 *     VMN_ERR_DEVICE and nothing else can be called.
 */
#ifndef VMNHIP_H
#define VMNHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum vmn_status {
    VMN_OK = 0,
    VMN_ERR_ARG = -1,       /* null/invalid handle, size mismatch, unsupported modulus size */
    VMN_ERR_DEVICE = -2,    /* HIP error (no device, launch failure); see vmn_last_error() */
    VMN_ERR_NOMEM = -3,     /* device or host allocation failed */
    VMN_ERR_FORMAT = -4,    /* value out of range on import (>= modulus) */
    VMN_ERR_UNSUPPORTED = -5
} vmn_status;

typedef struct vmn_ctx vmn_ctx;         /* one GPU + one stream + scratch workspace            */
typedef struct vmn_group vmn_group;     /* ModPGroup: subgroup of order q of Z_p^*, generator g */
typedef struct vmn_garray vmn_garray;   /* PGroupElementArray over a vmn_group (device)         */
typedef struct vmn_rarray vmn_rarray;   /* PRingElementArray / PFieldElementArray over Z_q       */

/* ---- library / context ------------------------------------------------------------------- */

const char* vmn_version(void);
/* Thread-local description of the last failing call in this thread (never NULL). */
const char* vmn_last_error(void);
/* for libraries layered on this ABI (vmnproofs.h): record the message vmn_last_error() returns */
void vmn_report_error(const char* message);

/* device: HIP device ordinal.  Fails with VMN_ERR_DEVICE when no gfx950 GPU is present. */
int vmn_ctx_create(int device, vmn_ctx** out);
void vmn_ctx_destroy(vmn_ctx* ctx);
/* Use an externally owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = the
 * context's own stream. */
int vmn_ctx_set_stream(vmn_ctx* ctx, void* hip_stream);
void* vmn_ctx_get_stream(vmn_ctx* ctx);
int vmn_ctx_synchronize(vmn_ctx* ctx);
/* The helper thread.  The reference runs ONE helper thread beside a party's protocol thread: it multiplies and permutes
 * the next input while the protocol thread verifies a proof (ref: P/mixnet/ShufflerElGamalSession.java:839-859, 894-944),
 * or writes byte trees (P/hvzk/CCPoSW.java:114-123).  A thread that calls vmn_ctx_helper_begin(ctx) becomes that helper
 * until vmn_ctx_helper_end(ctx): its calls on arrays / groups of ctx run on a second, high-priority stream with its own
 * pool, scratch and lock, so they neither wait for the protocol thread's calls nor queue behind its kernels (the long
 * fixed-base launches are one workgroup per tile so that slots free up every few milliseconds).
 * Ordering.  vmn_ctx_helper_mark (either thread) records "everything the protocol thread has queued up to here"; begin()
 * and helper_sync() order the helper's stream behind the LATEST mark (begin() sets one if there has never been any) --
 * not behind work queued after it, which is what lets the helper run beside that work.  The protocol thread marks right
 * after queueing what the helper needs (typically just before it starts the thread); end() waits for the helper's work,
 * so the protocol thread may use its results after joining the thread.  An array must not be freed by one thread while
 * the other still uses it -- the reference's rule for its own arrays.  One helper per context. */
int vmn_ctx_helper_mark(vmn_ctx* ctx);
int vmn_ctx_helper_begin(vmn_ctx* ctx);
int vmn_ctx_helper_sync(vmn_ctx* ctx);
int vmn_ctx_helper_end(vmn_ctx* ctx);
/* Number of compute units of the context's device (used by the benchmark to state the roofline). */
int vmn_ctx_num_cus(vmn_ctx* ctx);
/* Small arrays.  One element per lane fills the device only from ~1.3 x 10^5 elements on; real elections (the
 * reference's demo: 10^4 ciphertexts, BASELINE.json configs[0]) are far below.  For 2048-bit moduli every launch over at
 * most `items` elements therefore runs in the WIDE geometry -- the same rows, same results, four lanes per element, so
 * that each chain of dependent products is ~2.5 times shorter.  Default 40960 (the measured crossover; env
 * VMN_WIDE_MAX overrides it); 0 = never, SIZE_MAX = always.  A tuning knob: results never depend on it. */
int vmn_ctx_set_small_array_threshold(vmn_ctx* ctx, size_t items);
/* The same one step further: launches over at most `items` elements of a 2048-bit modulus spread every element over
 * EIGHT lanes (default 6144 elements, env VMN_WIDE8_MAX; 0 = never).  Takes precedence over the threshold above. */
int vmn_ctx_set_tiny_array_threshold(vmn_ctx* ctx, size_t items);
/* Memory accounting (operations / leak hunting): bytes and blocks of freed arrays cached for reuse, bytes of live
 * allocations handed out and not yet freed (arrays + temporaries), and a group's cached fixed-base tables. */
int vmn_ctx_memory_stats(vmn_ctx* ctx, size_t* pool_bytes, size_t* pool_blocks, size_t* live_bytes);
size_t vmn_group_table_bytes(const vmn_group* grp);

/* ---- groups -------------------------------------------------------------------------------
 * ModPGroup(p, q, g): ref: P/elgamal/ProtocolElGamal.java:738-800 (group shapes), the marshalled
 * example at demo/mixnet/benchmarks/bench_config:43.  p, q, g are big-endian, nbytes each.
 * Supported modulus sizes: 512, 1024, 2048, 3072, 4096 bits, and -- for completeness, untuned (the reference offers safe
 * primes up to 15 424 bits and benchmarks with a 15 492-bit group) -- 8192 and 16384 bits (a modulus runs in the smallest
 * size that holds it). */
int vmn_modp_group_create(vmn_ctx* ctx, const uint8_t* p_be, const uint8_t* q_be, const uint8_t* g_be,
                          size_t nbytes, vmn_group** out);
/* ECqPGroup over a named NIST prime curve: "P-256" and "P-384" -- the two north_star names, with the field primes compiled
 * into the kernels (csrc/ec_kernels.h FieldPrime) -- and, untuned, "P-224" (the curve the reference runs its own `check` on,
 * demo/mixnet/.checkbaseconf:59; square roots by Tonelli-Shanks, p = 1 mod 4) and "P-521" (benchmarks/bench_config:37).  Any
 * other name is VMN_ERR_UNSUPPORTED: the kernels assume a = -3 (dbl-2001-b), so the brainpool curves the reference also
 * offers (demo/mixnet/.conf:150-156) would need a general `a`, and P-192 is not instantiated.
 * ref: the default group of the reference, demo/mixnet/.conf:153 (P-256); SURVEY.md §2.3 K11.  Group elements cross the boundary as
 * x || y (elem_bytes = 2 * coordinate width, big-endian; the point at infinity is all 0xff bytes);
 * exponents are residues mod the group order.  Every vmn_garray_* call works on such groups with the
 * group operation = point addition ("mul") and exponentiation = scalar multiplication ("exp"). */
int vmn_ec_group_create(vmn_ctx* ctx, const char* curve_name, vmn_group** out);
void vmn_group_destroy(vmn_group* grp);
size_t vmn_group_elem_bytes(const vmn_group* grp);     /* bytes per group element on the wire */
size_t vmn_group_exp_bytes(const vmn_group* grp);      /* bytes per exponent (ring element) on the wire */
/* The two widths are those the group was created with until this is called (vmn_modp_group_create: nbytes for both;
 * curves: the coordinate size).  0 selects the reference's own width, the length of Java's
 * BigInteger.toByteArray() of the modulus resp. the order -- floor(bits / 8) + 1, e.g. 257-byte elements and 256-byte
 * exponents for an RFC 3526 2048-bit group, 33 / 33 for P-256 -- which is what VCR writes into byte-tree leaves
 * (the 15 492-bit group of demo/mixnet/benchmarks/bench_config:43 has 1 937-byte leaves).  A JNI
 * binding calls this with (0, 0) right after creating the group.  Must precede any other use of the group. */
int vmn_group_set_wire_bytes(vmn_group* grp, size_t elem_bytes, size_t exp_bytes);
/* PGroup accessors used by host-side protocol code: getElementOrder() (P/hvzk/PoSBasicTW.java:470 uses its
 * bit length), getg() (P/mixnet/PermutationCommitment.java:200), and the modulus / field prime.
 * kind: 0 = ModPGroup, 1 = ECqPGroup.  order / modulus: exp_bytes big-endian bytes; generator: elem_bytes. */
int vmn_group_kind(const vmn_group* grp);
vmn_ctx* vmn_group_ctx(const vmn_group* grp);        /* the context (main lane) the group was created in */
int vmn_group_get_order(const vmn_group* grp, uint8_t* q_be);
int vmn_group_get_modulus(const vmn_group* grp, uint8_t* p_be);
int vmn_group_get_generator(const vmn_group* grp, uint8_t* g_be);

/* ---- group element arrays (PGroupElementArray) --------------------------------------------- */

/* pGroup.toElementArray(size, reader) / unsafeToElementArray: import n fixed-width big-endian
 * values.  ref: P/hvzk/PoSBasicTW.java:507, 787-792; P/mixnet/ShufflerElGamalSession.java:205.
 * *all_in_range (may be NULL) is set to 0 if some value is >= p or == 0 (the array is then
 * still created, offending entries replaced by 1, mirroring the reference's "replace by trivial
 * value" convention); subgroup membership proper is vmn_garray_is_member. */
int vmn_garray_from_be(vmn_group* grp, const uint8_t* be, size_t n, vmn_garray** out, int* all_in_range);
/* array.toByteTree() payload: n * elem_bytes big-endian bytes.  ref: 174 toByteTree call sites, e.g.
 * P/hvzk/PoSBasicTW.java:694-699. */
int vmn_garray_to_be(const vmn_garray* a, uint8_t* be_out);
/* The same in the reference's byte-tree framing: node(N leaves of elem_bytes) =
 * 00 | uint32_be(N) | N x (01 | uint32_be(elem_bytes) | value); over a curve group N x node(leaf(x), leaf(y)), the point at
 * infinity as two leaves of 0xff bytes (= -1) [NOT-IN-REF: VCR's form, from the verifier specification].  Headers are written / checked on the
 * GPU.  from_bytetree: expected_n = 0 accepts any size (pGroup.toElementArray(0, reader)); *format_ok = 0
 * (and no array) when the tree is not a node of leaves of the right width (the reference's
 * ArithmFormatException / EIOException, which its callers catch: P/hvzk/PoSBasicTW.java:505-513). */
size_t vmn_garray_bytetree_size(const vmn_garray* a);
int vmn_garray_to_bytetree(const vmn_garray* a, uint8_t* out);
int vmn_garray_from_bytetree(vmn_group* grp, const uint8_t* bt, size_t len, size_t expected_n, vmn_garray** out,
                             int* format_ok, int* all_in_range);
size_t vmn_rarray_bytetree_size(const vmn_rarray* a);
int vmn_rarray_to_bytetree(const vmn_rarray* a, uint8_t* out);
int vmn_rarray_from_bytetree(vmn_group* grp, const uint8_t* bt, size_t len, size_t expected_n, vmn_rarray** out,
                             int* format_ok, int* all_in_range);
size_t vmn_garray_size(const vmn_garray* a);
/* the group an array belongs to (PGroupElementArray.getPGroup(), PRingElementArray.getPRing()): a binding sizes the host
 * buffers of vmn_*_to_be / _get / _prod ... from it (jni/: every byte[] is checked against the bytes the callee touches) */
vmn_group* vmn_garray_group(const vmn_garray* a);
vmn_group* vmn_rarray_group(const vmn_rarray* a);
void vmn_garray_free(vmn_garray* a);                   /* PGroupElementArray.free() */

/* K1a  X.exp(E): out[i] = X[i]^E[i].  ref: P/hvzk/PoSBasicTW.java:1032; P/hvzk/PoSCBasicTW.java:694.
 * ebits = bit length bound of the exponents actually used (<= 8*exp_bytes; 0 = full width). */
int vmn_garray_exp_array(const vmn_garray* x, const vmn_rarray* e, int ebits, vmn_garray** out);
/* Same with integer exponents that are NOT reduced mod q (LargeIntegerArray, e.g. the
 * n_e+n_v+n_r-bit k_E):  exps_be = n big-endian values of ebytes each. */
int vmn_garray_exp_ints(const vmn_garray* x, const uint8_t* exps_be, size_t ebytes, int ebits, vmn_garray** out);
/* K1b  X.exp(e) with one shared exponent.  ref: P/hvzk/PoSBasicTW.java:1028;
 * P/mixnet/ShufflerElGamalSession.java:506; P/mixnet/PermutationCommitment.java:357;
 * P/elgamal/DistrElGamalSession.java:384-385. */
int vmn_garray_exp_scalar(const vmn_garray* x, const uint8_t* e_be, size_t ebytes, vmn_garray** out);
/* K2  g.exp(E): out[i] = base^E[i] for one fixed base.  ref: P/mixnet/ShufflerElGamalSession.java:407,
 * 658; P/hvzk/PoSBasicTW.java:447, 606, 608, 644, 646, 1030; P/mixnet/PermutationCommitment.java:200. */
int vmn_group_exp_fixed(vmn_group* grp, const uint8_t* base_be, const vmn_rarray* e, vmn_garray** out);
/* Session setup: build the fixed-base table of a long-lived base (the generator, the public key) for arrays of about
 * n_hint exponents and about uses_hint calls.  The window is the one that is best over those uses (w = 19 = a 17 GB
 * table at 2048 bits, N = 10^6, instead of w = 16 = 2.5 GB for a base seen once: 108 instead of 128 products per
 * exponentiation); allocating it takes ~0.6 s, which is why it belongs to setup.  VCR precomputes fixed-base tables
 * the same way when it is handed a generator it will exponentiate many times.  Bases that are never announced get
 * the one-call window on first use. */
int vmn_group_precompute_fixed(vmn_group* grp, const uint8_t* base_be, size_t n_hint, int uses_hint);
/* The table of a base that will not be used again (a prover's per-proof base h_0 when the proof object is freed -- the
 * reference frees a proof's arrays in PoSBasicTW.free(), P/hvzk/PoSBasicTW.java:1088-1101; VCR keeps no table beyond a call):
 * it leaves the group's cache and its memory serves the next table of that size.  Unknown base: no effect. */
int vmn_group_release_fixed(vmn_group* grp, const uint8_t* base_be);
/* K3  X.expProd(E) = prod_i X[i]^E[i] -> one element (big-endian, elem_bytes).
 * ref: P/hvzk/PoSBasicTW.java:408, 409, 481, 690, 1021, 1063; P/hvzk/CCPoSBasicW.java:380, 391, 497-503. */
int vmn_garray_expprod(const vmn_garray* x, const vmn_rarray* e, int ebits, uint8_t* out_be);
int vmn_garray_expprod_ints(const vmn_garray* x, const uint8_t* exps_be, size_t ebytes, int ebits, uint8_t* out_be);
/* The same for k arrays under ONE exponent array (the 2*width component arrays of a ciphertext array:
 * PPGroupElementArray.expProd, ref: P/hvzk/PoSBasicTW.java:409, 690, 1063; P/hvzk/CCPoSBasicW.java:391, 498, 567):
 * the exponent digits are sorted once; out_be receives k elements. */
int vmn_garray_expprod_multi(const vmn_garray* const* xs, size_t k, const vmn_rarray* e, int ebits, uint8_t* out_be);
/* The same in two halves.  _begin queues the device part and returns a handle; vmn_pending_finish waits for it, completes
 * the k elements into out_be and releases the handle (also when it fails); vmn_pending_free abandons a handle that is not
 * finished.  Between the two the caller queues its other device work: the tail of a multi-exponentiation over a modular
 * group is a chain of squarings on the host (one per exponent bit), during which the device would otherwise have nothing
 * to do -- at the reference's demo size (10^4 ciphertexts) a tenth of a proof.  The arrays must stay alive until finish. */
typedef struct vmn_pending vmn_pending;
int vmn_garray_expprod_multi_begin(const vmn_garray* const* xs, size_t k, const vmn_rarray* e, int ebits, vmn_pending** out);
int vmn_pending_finish(vmn_pending* p, uint8_t* out_be);
size_t vmn_pending_bytes(const vmn_pending* p);   /* what finish writes: k elements */
void vmn_pending_free(vmn_pending* p);
/* K4  X.mul(Y).  ref: P/mixnet/ShufflerElGamalSession.java:273, 789, 850 (the re-encryption);
 * P/hvzk/PoSBasicTW.java:448, 610, 648, 1029, 1033. */
int vmn_garray_mul(const vmn_garray* x, const vmn_garray* y, vmn_garray** out);
/* Element-wise inverse X.inv() by Montgomery's batch inversion (two product scans + one host inversion
 * of the total + 2N products instead of N exponentiations).  Serves the negative modified Lagrange
 * coefficients of pGroup.expProd(bases[], integers[], bitLength),
 * ref: P/elgamal/DistrElGamalSessionBasic.java:487-502 (K3'), :406-452 (coefficients may be negative). */
int vmn_garray_inv(const vmn_garray* x, vmn_garray** out);
/* out[i] = x[i]^e * y[i]^f[i] as ONE simultaneous power: the squarings are shared between the two exponents
 * (max(bits(e), fbits) of them instead of the sum).  Curves: one chain of doublings for the two scalar multiplications
 * (e acts through its residue modulo the group order).  The
 * verifiers' check (B) of PoSBasicTW.java:1023-1042 in the form B_i^v (B_{i-1}^{-1})^{k_E,i} B'_i = g^{k_B,i}. */
int vmn_garray_exp2(const vmn_garray* x, const uint8_t* e_be, size_t ebytes, const vmn_garray* y, const vmn_rarray* f, int fbits,
                    vmn_garray** out);
/* out_x[i] = x[i]^e and out_y[i] = y[i]^f[i], the two powers of the same check in its separate form
 * (PoSBasicTW.java:1028-1033: B.exp(v) and B_shift.exp(k_E)), as ONE launch when the arrays are too small to fill
 * the device one after the other; otherwise (and over curves) exactly vmn_garray_exp_scalar + vmn_garray_exp_array.
 * x and y may differ in length; f has y's. */
int vmn_garray_exp_pair(const vmn_garray* x, const uint8_t* e_be, size_t ebytes, const vmn_garray* y, const vmn_rarray* f, int fbits,
                        vmn_garray** out_x, vmn_garray** out_y);
/* K5  X.prod() -> one element.  ref: P/hvzk/PoSBasicTW.java:1013; P/hvzk/PoSCBasicTW.java:667. */
int vmn_garray_prod(const vmn_garray* x, uint8_t* out_be);
/* K6  X.equals(Y).  ref: P/hvzk/PoSBasicTW.java:1035; P/hvzk/PoSCBasicTW.java:697. */
int vmn_garray_equals(const vmn_garray* x, const vmn_garray* y, int* equal);
/* K7  data movement.  permute: out[i] = X[perm[i]] (gather; see SURVEY.md App. B on the convention);
 * shiftPush: (el, X[0..n-2]); copyOfRange [from, to); extract keeps X[i] where keep[i] != 0.
 * ref: P/mixnet/ShufflerElGamalSession.java:278, 684-703, 792; P/hvzk/PoSBasicTW.java:451, 637-638, 1031;
 * P/mixnet/PermutationCommitment.java:398-405, 462-469. */
int vmn_garray_permute(const vmn_garray* x, const uint32_t* perm_host, vmn_garray** out);
/* General gather: out[i] = X[idx[i]], i < n_out (n_out may differ from the array's size: a shard of a
 * permuted array is a gather of the replicated input). */
int vmn_garray_gather(const vmn_garray* x, const uint32_t* idx_host, size_t n_out, vmn_garray** out);
int vmn_garray_shift_push(const vmn_garray* x, const uint8_t* el_be, vmn_garray** out);
int vmn_garray_copy_range(const vmn_garray* x, size_t from, size_t to, vmn_garray** out);
int vmn_garray_extract(const vmn_garray* x, const uint8_t* keep_host, vmn_garray** out);
int vmn_garray_get(const vmn_garray* x, size_t i, uint8_t* out_be);
/* Subgroup membership of every element; *all_members = 1/0.  Part of K10 (the check pGroup.toElementArray makes
 * when an array is read).  Safe-prime groups up to 3072 bits: Jacobi symbol (x / p) = 1 by the binary algorithm,
 * one element per lane (10^6 elements in 46 ms); otherwise x^q = 1 (749 ms); curves: the import's on-curve check
 * already is the membership test (cofactor 1). */
int vmn_garray_is_member(const vmn_garray* x, int* all_members);

/* ---- ring element arrays over Z_q (PRingElementArray / PFieldElementArray) ------------------ */

int vmn_rarray_from_be(vmn_group* grp, const uint8_t* be, size_t n, vmn_rarray** out, int* all_in_range);
int vmn_rarray_to_be(const vmn_rarray* a, uint8_t* be_out);
size_t vmn_rarray_size(const vmn_rarray* a);
void vmn_rarray_free(vmn_rarray* a);
/* K8.  ref: P/hvzk/PoSBasicTW.java:596 (recLin), 604 (prods), 642-645, 861-863, 873-878, 1014;
 * P/hvzk/PoSCBasicTW.java:410, 418, 483-486, 612-613, 623-627, 668; P/hvzk/CCPoSBasicW.java:467-478. */
int vmn_rarray_mul(const vmn_rarray* x, const vmn_rarray* y, vmn_rarray** out);
int vmn_rarray_add(const vmn_rarray* x, const vmn_rarray* y, vmn_rarray** out);
int vmn_rarray_neg(const vmn_rarray* x, vmn_rarray** out);
/* x.mulAdd(v, y): out[i] = x[i]*v + y[i] with scalar v (big-endian, exp_bytes); y == NULL: out[i] = x[i]*v. */
int vmn_rarray_mul_add(const vmn_rarray* x, const uint8_t* v_be, const vmn_rarray* y, vmn_rarray** out);
/* b.recLin(e): x[0] = b[0], x[i] = x[i-1]*e[i] + b[i]; last = x[n-1] (may be NULL). */
int vmn_rarray_rec_lin(const vmn_rarray* b, const vmn_rarray* e, vmn_rarray** out_x, uint8_t* last_be);
/* e.prods(): y[i] = prod_{j<=i} e[j]. */
int vmn_rarray_prods(const vmn_rarray* e, vmn_rarray** out);
int vmn_rarray_inner_product(const vmn_rarray* x, const vmn_rarray* y, uint8_t* out_be);
int vmn_rarray_sum(const vmn_rarray* x, uint8_t* out_be);
/* k of those in ONE round trip: out_be[i] = <xs[i], ys[i]>, or the sum of xs[i] where ys[i] is NULL (a reply's <r, e'>,
 * sum r and <s_c, e>, ref: P/hvzk/PoSBasicTW.java:856-888).  out_be receives k ring elements. */
int vmn_rarray_inner_products(const vmn_rarray* const* xs, const vmn_rarray* const* ys, size_t k, uint8_t* out_be);
int vmn_rarray_prod(const vmn_rarray* x, uint8_t* out_be);
int vmn_rarray_permute(const vmn_rarray* x, const uint32_t* perm_host, vmn_rarray** out);
int vmn_rarray_gather(const vmn_rarray* x, const uint32_t* idx_host, size_t n_out, vmn_rarray** out);
int vmn_rarray_shift_push(const vmn_rarray* x, const uint8_t* el_be, vmn_rarray** out);
int vmn_rarray_equals(const vmn_rarray* x, const vmn_rarray* y, int* equal);
int vmn_rarray_get(const vmn_rarray* x, size_t i, uint8_t* out_be);
int vmn_rarray_copy_range(const vmn_rarray* x, size_t from, size_t to, vmn_rarray** out);
/* Largest bit length among the entries (0 for an empty or all-zero array).  A verifier must use EVERY bit of an
 * exponent array it received (pField.toElementArray parses full field elements, ref: P/hvzk/PoSBasicTW.java:985-989, and
 * h.expProd(k_E) :1021, B_shift.exp(k_E) :1032 use them whole); this tells it how many there are, so that honest
 * replies (n_e + n_v + n_r + 1 bits) keep their short exponentiations and anything longer is still computed exactly. */
int vmn_rarray_max_bits(const vmn_rarray* x, int* bits);

/* ---- pseudo-random derivations ---------------------------------------------
 * VCR's PRGHeuristic and RandomOracle over SHA-256 / SHA-384 / SHA-512 (the three the reference offers,
 * ref: P/elgamal/ProtocolElGamal.java:352-371 PRG, :413-434 random-oracle hash), restated from their published definition
 * (the classes are not in the reference tree): PRG(seed) = H(seed || uint32_be(0)) || H(seed || uint32_be(1)) || ...;
 * RO_nout(d) = first ceil(nout/8) bytes of PRG(H(uint32_be(nout) || d)) with the superfluous leading bits cleared.
 * A PRG seed has the digest's length (PRGHeuristic.minNoSeedBytes), so seedlen = 32 / 48 / 64 selects the hash everywhere
 * a seed is taken.  The SHA-256 constructions are pinned by the published known-answer vectors, SHA-384 / SHA-512 by
 * hashlib (tests/test_prg.py). */
int vmn_prg_bytes(const uint8_t* seed, size_t seedlen, uint8_t* out, size_t nbytes);              /* host */
int vmn_random_oracle(const uint8_t* data, size_t len, int nout_bits, uint8_t* out);            /* host, SHA-256 */
int vmn_random_oracle_hash(int hash_bits, const uint8_t* data, size_t len, int nout_bits, uint8_t* out);   /* 256 / 384 / 512 */
/* The random vector of a proof, generated on the device: prg.setSeed(seed); LargeIntegerArray.random(n, bits, prg)
 * as field elements.  ref: P/hvzk/PoSBasicTW.java:533-538, PoSCBasicTW.java:350-355, CCPoSBasicW.java:330-335.
 * Value i = the i-th ceil(bits/8) bytes of the stream, leading bits cleared; reduced mod q when it can reach q. */
int vmn_rarray_from_prg(vmn_group* grp, const uint8_t* seed, size_t seedlen, size_t n, int bits, vmn_rarray** out);
/* Parts of the same array without the rest (the stream is counter mode): values [first, first + n), resp. the values
 * idx[0 .. n-1] (host indices) -- equal to copy_range / gather of vmn_rarray_from_prg(.., N, ..).  A rank of a sharded
 * proof generates its own positions of r, s, b, beta, epsilon, e and the rows e_{pi^-1(i)}, r_{pi(i)}, s_{pi^-1(i)} it
 * reads through the permutation, instead of all N values on every GPU (the reference has no counterpart:
 * P/hvzk/PoSBasicTW.java:446, 473, 552-554 run on one machine). */
int vmn_rarray_from_prg_range(vmn_group* grp, const uint8_t* seed, size_t seedlen, size_t first, size_t n, int bits, vmn_rarray** out);
int vmn_rarray_from_prg_gather(vmn_group* grp, const uint8_t* seed, size_t seedlen, const uint32_t* idx, size_t n, int bits,
                               vmn_rarray** out);
/* Independent generators: pGroup.randomElementArray(n, prg, rbitlen) of a ModPGroup, generated on the
 * device.  ref: P/distr/IndependentGeneratorsRO.java:117-130 (seed = RO(globalPrefix || bytetree(sid))).
 * t_i = the i-th ceil((bits(p) + rbitlen)/8) bytes, leading bits cleared; h_i = t_i^((p-1)/q) mod p -- a squaring for
 * the safe-prime groups the reference generates, a power with the group's own cofactor otherwise (the reference's
 * ModPGroup_1024_256, demo/mixnet/group_descriptions:29; tests/test_bytetree.py).
 * ECqPGroup (P-256 is the reference's default group, demo/mixnet/.conf:153): the values are candidates for x = t mod p,
 * kept when x^3 - 3x + b is a square, with the smaller root as y; element i is the i-th kept candidate (candidates are
 * tested in parallel, the kept ones compacted in order).
 * The derivations follow the specification's text; they are not pinned by a vector of the reference [NOT-IN-REF]. */
int vmn_garray_from_prg(vmn_group* grp, const uint8_t* seed, size_t seedlen, size_t n, int rbitlen, vmn_garray** out);

/* ---- partial results for multi-GPU sharding --------------------------------
 * Each rank holds a contiguous shard; expProd/prod partials are single elements that the host
 * side exchanges (RCCL all-gather of G x elem_bytes) and multiplies.  This multiplies k partials
 * given as k*elem_bytes big-endian bytes into one element. */
int vmn_group_mul_partials(vmn_group* grp, const uint8_t* partials_be, size_t k, uint8_t* out_be);

/* ---- instrumentation ---------------------------------------------------------------------- */
/* Name, launch count and accumulated device time (ms, HIP events on the context stream) of the
 * kernel families since the last reset; used by bench.py for the roofline line.  Timing is off
 * by default (no events recorded). */
int vmn_ctx_timing_enable(vmn_ctx* ctx, int on);
int vmn_ctx_timing_reset(vmn_ctx* ctx);
/* family: "modpow", "modmul", "fixed", "expprod", ...; returns launches and total ms. */
int vmn_ctx_timing_get(vmn_ctx* ctx, const char* family, long* launches, double* total_ms);
/* All families as text lines "family launches total_ms executed_mads canonical_macs\n" into buf (truncated to len-1 bytes):
 * executed_mads = v_mad_u64_u32 multiply-adds of the 28-bit-limb kernels, canonical_macs = the same products priced at
 * SURVEY.md 8d's M(s) = 2 s^2 + s / Q(s) on s = bits / 32 limbs (the unit of the headline roofline). */
int vmn_ctx_timing_report(vmn_ctx* ctx, char* buf, size_t len);

#ifdef __cplusplus
}
#endif
#endif /* VMNHIP_H */
