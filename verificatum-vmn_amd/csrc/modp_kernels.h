// modp_kernels.h — gfx950 kernels over arrays of residues in M28 form (see mont28.h).
//
// Device array layout ("AoS"): element i occupies W = stride_for_limbs(S) consecutive 32-bit
// words at base + i*W; limb j (28 bits) in word j, padding words zero.  Elements are contiguous
// so that gathers / permutations / sharding by contiguous ranges move whole 16-byte-aligned
// rows, and one lane streams its element with 16-byte loads.
//
// Every arithmetic kernel is "one element per lane": the lane's multiplicand sits in VGPRs, its
// multiplier is staged in LDS as lds[limb*256 + tid] (bank = tid mod 32: conflict-free
// ds_read_b32 / ds_write_b32), the modulus limbs are wave-uniform SGPRs.
#pragma once
#include "mont28.h"

namespace vmn {

constexpr int BLOCK = 256;   // threads per workgroup: 4 waves, one per SIMD

// ---------------------------------------------------------------------------------------------
// element movement helpers
// ---------------------------------------------------------------------------------------------
template <int S>
__device__ __forceinline__ void load_elem(u32 (&a)[S], const u32* __restrict__ p) {
    constexpr int W = stride_for_limbs(S);
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int k = 0; k < W / 4; ++k) {
        uint4 v = q[k];
        if (4 * k + 0 < S) a[4 * k + 0] = v.x;
        if (4 * k + 1 < S) a[4 * k + 1] = v.y;
        if (4 * k + 2 < S) a[4 * k + 2] = v.z;
        if (4 * k + 3 < S) a[4 * k + 3] = v.w;
    }
}
template <int S>
__device__ __forceinline__ void store_elem(u32* __restrict__ p, const u32 (&a)[S]) {
    constexpr int W = stride_for_limbs(S);
    uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int k = 0; k < W / 4; ++k) {
        uint4 v;
        v.x = 4 * k + 0 < S ? a[4 * k + 0] : 0;
        v.y = 4 * k + 1 < S ? a[4 * k + 1] : 0;
        v.z = 4 * k + 2 < S ? a[4 * k + 2] : 0;
        v.w = 4 * k + 3 < S ? a[4 * k + 3] : 0;
        q[k] = v;
    }
}
// global element -> this lane's LDS column (multiplier operand)
template <int S>
__device__ __forceinline__ void load_elem_to_lds(u32* bl, const u32* __restrict__ p) {
    constexpr int W = stride_for_limbs(S);
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int k = 0; k < W / 4; ++k) {
        uint4 v = q[k];
        if (4 * k + 0 < S) bl[(4 * k + 0) * BLOCK] = v.x;
        if (4 * k + 1 < S) bl[(4 * k + 1) * BLOCK] = v.y;
        if (4 * k + 2 < S) bl[(4 * k + 2) * BLOCK] = v.z;
        if (4 * k + 3 < S) bl[(4 * k + 3) * BLOCK] = v.w;
    }
}
template <int S>
__device__ __forceinline__ void regs_to_lds(u32* bl, const u32 (&a)[S]) {
#pragma unroll
    for (int j = 0; j < S; ++j) bl[j * BLOCK] = a[j];
}
template <int S>
__device__ __forceinline__ void load_modulus(u32 (&n)[S], const u32* __restrict__ nmod) {
#pragma unroll
    for (int j = 0; j < S; ++j) n[j] = nmod[j];     // uniform address: scalar loads into SGPRs
}

// r = a * (lane's LDS column) / R mod N, limbs normalised, value < 2N
template <int S>
__device__ __forceinline__ void mont_mul(u32 (&r)[S], const u32 (&a)[S], const u32* bl, const u32 (&n)[S], u32 n0inv) {
    u64 T[S];
    mont_mul_columns<S>(T, a, bl, BLOCK, n, n0inv);
    normalize_columns<S>(r, T);
}

// r = a^2 / R mod N; the lane's LDS column must hold a copy of a
template <int S>
__device__ __forceinline__ void mont_sqr(u32 (&r)[S], const u32 (&a)[S], const u32* bl, const u32 (&n)[S], u32 n0inv) {
    u64 T[S];
    mont_sqr_columns<S>(T, a, bl, BLOCK, n, n0inv);
    normalize_columns<S>(r, T);
}

// ---------------------------------------------------------------------------------------------
// radix conversion between packed 32-bit words (NW words, little-endian) and 28-bit limbs
// ---------------------------------------------------------------------------------------------
template <int S, int NW>
__device__ __forceinline__ void words_to_limbs(u32 (&l)[S], const u32 (&w)[NW]) {
#pragma unroll
    for (int j = 0; j < S; ++j) {
        const int bit = 28 * j, k = bit / 32, sh = bit % 32;
        u32 lo = k < NW ? w[k] : 0, hi = k + 1 < NW ? w[k + 1] : 0;
        u64 both = ((u64)hi << 32) | lo;
        l[j] = (u32)(both >> sh) & LIMB_MASK;
    }
}
template <int S, int NW>
__device__ __forceinline__ void limbs_to_words(u32 (&w)[NW], const u32 (&l)[S]) {
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const int bit = 32 * k, j = bit / 28, sh = bit % 28;      // word k starts inside limb j
        u64 acc = 0;
        if (j < S) acc = (u64)l[j] >> sh;
        if (j + 1 < S) acc |= (u64)l[j + 1] << (28 - sh);
        if (j + 2 < S) acc |= (u64)l[j + 2] << (56 - sh);
        w[k] = (u32)acc;
    }
}

// ---------------------------------------------------------------------------------------------
// import / export:  big-endian fixed-width bytes  <->  M28 form
// ---------------------------------------------------------------------------------------------
// Word k (little-endian significance) of a big-endian integer of nbytes bytes at p; bytes beyond
// the integer read as zero.  nbytes need not be a multiple of 4 (Java's BigInteger encoding of a
// 2048-bit modulus is 257 bytes wide): whole words use one unaligned 4-byte load.
__device__ __forceinline__ u32 load_be_word(const uint8_t* __restrict__ p, long nbytes, int k) {
    long off = nbytes - 4L * (k + 1);
    if (off >= 0) {
        u32 v;
        __builtin_memcpy(&v, p + off, 4);
        return __builtin_bswap32(v);
    }
    u32 v = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        long o = nbytes - 1 - (4L * k + b);
        if (o >= 0) v |= (u32)p[o] << (8 * b);
    }
    return v;
}
__device__ __forceinline__ void store_be_word(uint8_t* __restrict__ p, long nbytes, int k, u32 w) {
    long off = nbytes - 4L * (k + 1);
    if (off >= 0) {
        u32 v = __builtin_bswap32(w);
        __builtin_memcpy(p + off, &v, 4);
        return;
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        long o = nbytes - 1 - (4L * k + b);
        if (o >= 0) p[o] = (uint8_t)(w >> (8 * b));
    }
}

// flags[0] |= 1 if some value >= N ; flags[0] |= 2 if some value == 0 (only reported).
// Out-of-range values are replaced by `one` (x = 1) -- the reference substitutes trivial values
// for malformed input (P/hvzk/PoSBasicTW.java:794-815).
template <int S, int NW>
__global__ void __launch_bounds__(BLOCK, (S <= 74 ? 2 : 1))
k_import_be(u32* __restrict__ out, const uint8_t* __restrict__ be, size_t nbytes, size_t n,
            const u32* __restrict__ nmod, u32 n0inv, const u32* __restrict__ rr, u32* __restrict__ flags) {
    constexpr int W = stride_for_limbs(S);
    extern __shared__ u32 lds[];
    u32* bl = lds + threadIdx.x;
    u32 nn[S];
    load_modulus<S>(nn, nmod);
    size_t el = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    bool live = el < n;
    size_t ec = live ? el : n - 1;
    const uint8_t* src = be + ec * nbytes;
    u32 w[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) w[k] = load_be_word(src, (long)nbytes, k);
    u32 extra = 0;                                    // leading bytes beyond NW words must be zero
    for (long o = (long)nbytes - 4L * NW - 1; o >= 0; --o) extra |= src[o];
    u32 a[S];
    words_to_limbs<S, NW>(a, w);
    // range check: a < N  (borrow of a - N)
    int32_t borrow = 0;
    u32 nz = 0;
#pragma unroll
    for (int j = 0; j < S; ++j) {
        int32_t v = (int32_t)a[j] - (int32_t)nn[j] + borrow;
        borrow = v >> LIMB_BITS;
        nz |= a[j];
    }
    bool bad = (borrow == 0) || extra != 0;     // a >= N
    if (live && bad) atomicOr(flags, 1u);
    if (live && nz == 0) atomicOr(flags, 2u);
    if (bad) {
#pragma unroll
        for (int j = 0; j < S; ++j) a[j] = j == 0 ? 1u : 0u;
    }
    // to Montgomery form: a * RR / R
#pragma unroll
    for (int j = 0; j < S; ++j) bl[j * BLOCK] = rr[j];
    u32 r[S];
    mont_mul<S>(r, a, bl, nn, n0inv);
    canonicalize<S>(r, nmod);
    if (live) store_elem<S>(out + el * W, r);
}

template <int S, int NW>
__global__ void __launch_bounds__(BLOCK, (S <= 74 ? 2 : 1))
k_export_be(uint8_t* __restrict__ be, size_t nbytes, const u32* __restrict__ in, size_t n,
            const u32* __restrict__ nmod, u32 n0inv) {
    constexpr int W = stride_for_limbs(S);
    extern __shared__ u32 lds[];
    u32* bl = lds + threadIdx.x;
    u32 nn[S];
    load_modulus<S>(nn, nmod);
    size_t el = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    bool live = el < n;
    size_t ec = live ? el : n - 1;
    u32 a[S];
    load_elem<S>(a, in + ec * W);
#pragma unroll
    for (int j = 0; j < S; ++j) bl[j * BLOCK] = j == 0 ? 1u : 0u;       // multiply by 1: leaves the Montgomery domain
    u32 r[S];
    mont_mul<S>(r, a, bl, nn, n0inv);
    canonicalize<S>(r, nmod);
    u32 w[NW];
    limbs_to_words<S, NW>(w, r);
    if (live) {
        uint8_t* dst = be + el * nbytes;
#pragma unroll
        for (int k = 0; k < NW; ++k) store_be_word(dst, (long)nbytes, k, w[k]);
        for (long o = (long)nbytes - 4L * NW - 1; o >= 0; --o) dst[o] = 0;
    }
}

// M28 form -> packed little-endian words of the standard representative (exponent use)
template <int S, int NW>
__global__ void __launch_bounds__(BLOCK, (S <= 74 ? 2 : 1))
k_to_words(u32* __restrict__ out, const u32* __restrict__ in, size_t n, const u32* __restrict__ nmod, u32 n0inv) {
    constexpr int W = stride_for_limbs(S);
    extern __shared__ u32 lds[];
    u32* bl = lds + threadIdx.x;
    u32 nn[S];
    load_modulus<S>(nn, nmod);
    size_t el = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    bool live = el < n;
    size_t ec = live ? el : n - 1;
    u32 a[S];
    load_elem<S>(a, in + ec * W);
#pragma unroll
    for (int j = 0; j < S; ++j) bl[j * BLOCK] = j == 0 ? 1u : 0u;
    u32 r[S];
    mont_mul<S>(r, a, bl, nn, n0inv);
    canonicalize<S>(r, nmod);
    u32 w[NW];
    limbs_to_words<S, NW>(w, r);
    if (live) {
        uint4* dst = reinterpret_cast<uint4*>(out + el * NW);
#pragma unroll
        for (int k = 0; k < NW / 4; ++k) dst[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
    }
}

// ---------------------------------------------------------------------------------------------
// K4: out[i] = x[i] * y[i]        (ystride = 0: every x[i] times the single element y)
// ---------------------------------------------------------------------------------------------
template <int S>
__global__ void __launch_bounds__(BLOCK, (S <= 74 ? 2 : 1))
k_mul(u32* __restrict__ out, const u32* __restrict__ x, const u32* __restrict__ y, size_t ystride, size_t n,
      const u32* __restrict__ nmod, u32 n0inv) {
    constexpr int W = stride_for_limbs(S);
    extern __shared__ u32 lds[];
    u32* bl = lds + threadIdx.x;
    u32 nn[S];
    load_modulus<S>(nn, nmod);
    size_t el = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    bool live = el < n;
    size_t ec = live ? el : n - 1;
    u32 a[S];
    load_elem<S>(a, x + ec * W);
    load_elem_to_lds<S>(bl, y + ec * ystride);
    u32 r[S];
    mont_mul<S>(r, a, bl, nn, n0inv);
    canonicalize<S>(r, nmod);
    if (live) store_elem<S>(out + el * W, r);
}

// ---------------------------------------------------------------------------------------------
// K1a / K1b: out[i] = x[i] ^ e[i]   fixed-window (wbits), left to right, per-lane exponents.
//   e: packed little-endian words, element i at e + i*estride (estride = 0: one shared exponent)
//   tab: scratch of gridDim.x*BLOCK*(2^wbits)*W words: the lane's table of x^0..x^(2^w-1), one
//        contiguous row per entry (the lane gathers its entry with 16-byte loads).
// Persistent grid: a workgroup loops over tiles of BLOCK elements.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 exp_digit(const u32* __restrict__ ep, int ewords, int pos, int wbits) {
    int k = pos >> 5, sh = pos & 31;
    u32 lo = k < ewords ? ep[k] : 0;
    u32 hi = k + 1 < ewords ? ep[k + 1] : 0;
    u64 both = ((u64)hi << 32) | lo;
    return (u32)(both >> sh) & ((1u << wbits) - 1);
}

template <int S>
__global__ void __launch_bounds__(BLOCK, (S <= 74 ? 2 : 1))
k_modpow(u32* __restrict__ out, const u32* __restrict__ x, const u32* __restrict__ e, int ewords, size_t estride,
         int ebits, int wbits, size_t n, const u32* __restrict__ nmod, u32 n0inv, const u32* __restrict__ one_m,
         u32* __restrict__ tab) {
    constexpr int W = stride_for_limbs(S);
    extern __shared__ u32 lds[];
    u32* bl = lds + threadIdx.x;
    u32 nn[S];
    load_modulus<S>(nn, nmod);
    const size_t ntiles = (n + BLOCK - 1) / BLOCK;
    const int tsize = 1 << wbits;
    u32* mytab = tab + ((size_t)blockIdx.x * BLOCK + threadIdx.x) * (size_t)tsize * W;
    const int nwin = (ebits + wbits - 1) / wbits;

    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        size_t el = t * BLOCK + threadIdx.x;
        bool live = el < n;
        size_t ec = live ? el : n - 1;
        const u32* ep = e + ec * estride;
        u32 a[S];
        // table: tab[0] = 1, tab[1] = x, tab[k] = tab[k-1] * x
        load_elem<S>(a, x + ec * W);
        {
            u32 o[S];
#pragma unroll
            for (int j = 0; j < S; ++j) o[j] = one_m[j];
            store_elem<S>(mytab, o);
        }
        store_elem<S>(mytab + W, a);
        regs_to_lds<S>(bl, a);
#pragma unroll 1
        for (int k = 2; k < tsize; ++k) {
            u32 r[S];
            mont_mul<S>(r, a, bl, nn, n0inv);          // x * tab[k-1]
            store_elem<S>(mytab + (size_t)k * W, r);
            regs_to_lds<S>(bl, r);
        }
        // main loop
        u32 d = exp_digit(ep, ewords, (nwin - 1) * wbits, wbits);
        load_elem<S>(a, mytab + (size_t)d * W);
#pragma unroll 1
        for (int wi = nwin - 2; wi >= 0; --wi) {
#pragma unroll 1
            for (int s = 0; s < wbits; ++s) {
                regs_to_lds<S>(bl, a);
                mont_sqr<S>(a, a, bl, nn, n0inv);
            }
            d = exp_digit(ep, ewords, wi * wbits, wbits);
            load_elem_to_lds<S>(bl, mytab + (size_t)d * W);
            mont_mul<S>(a, a, bl, nn, n0inv);
        }
        canonicalize<S>(a, nmod);
        if (live) store_elem<S>(out + el * W, a);
    }
}

}  // namespace vmn

// =============================================================================================
// second part: fixed-base tables (K2), multi-exponentiation (K3), reductions (K5), comparison
// (K6), data movement (K7) and the ring kernels over Z_q (K8).
// =============================================================================================
namespace vmn {

// ---------------------------------------------------------------------------------------------
// K6: flags[0] |= 1 if x != y anywhere.  One thread per 16-byte chunk (HBM-bound, coalesced).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLOCK) k_compare(const uint4* __restrict__ x, const uint4* __restrict__ y,
                                                   size_t nchunks, u32* __restrict__ flags) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    bool diff = false;
    for (; i < nchunks; i += (size_t)gridDim.x * BLOCK) {
        uint4 a = x[i], b = y[i];
        diff |= (a.x != b.x) | (a.y != b.y) | (a.z != b.z) | (a.w != b.w);
    }
    if (__any(diff) && (threadIdx.x & 63) == 0) atomicOr(flags, 1u);
}

// ---------------------------------------------------------------------------------------------
// K7: out[i] = in[idx[i]]  (idx[i] == 0xffffffff: out[i] = fill).  One thread per 16-byte chunk of
// a row: rows are contiguous W-word records, so a gather moves whole aligned rows.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLOCK) k_gather(uint4* __restrict__ out, const uint4* __restrict__ in,
                                                  const u32* __restrict__ idx, const uint4* __restrict__ fill,
                                                  size_t n_out, int chunks_per_row) {
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    size_t total = n_out * (size_t)chunks_per_row;
    for (; t < total; t += (size_t)gridDim.x * BLOCK) {
        size_t row = t / chunks_per_row;
        int c = (int)(t % chunks_per_row);
        u32 src = idx[row];
        out[t] = src == 0xffffffffu ? fill[c] : in[(size_t)src * chunks_per_row + c];
    }
}

// ---------------------------------------------------------------------------------------------
// K5 and friends: strided reduction.  out[seg][j] = OP_k x[seg][j + k*L], k = 0 .. ceil(len/L)-1,
// for j < L.  OP = Montgomery product (MUL = true) or modular sum.  nseg segments of `len`
// elements each; the output has nseg segments of L elements.
// ---------------------------------------------------------------------------------------------
template <int S>
__device__ __forceinline__ void mod_add(u32 (&r)[S], const u32 (&a)[S], const u32 (&b)[S], const u32* __restrict__ nmod) {
    u32 c = 0;
#pragma unroll
    for (int j = 0; j < S; ++j) {
        u32 v = a[j] + b[j] + c;
        r[j] = v & LIMB_MASK;
        c = v >> LIMB_BITS;
    }
    canonicalize<S>(r, nmod);      // a, b < N  =>  a + b < 2N
}

template <int S, bool MUL>
__global__ void __launch_bounds__(BLOCK, (S <= 74 ? 2 : 1))
k_reduce_strided(u32* __restrict__ out, const u32* __restrict__ x, size_t len, size_t L, size_t nseg,
                 const u32* __restrict__ nmod, u32 n0inv) {
    constexpr int W = stride_for_limbs(S);
    extern __shared__ u32 lds[];
    u32* bl = lds + threadIdx.x;
    u32 nn[S];
    load_modulus<S>(nn, nmod);
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    bool live = t < nseg * L;
    size_t tc = live ? t : nseg * L - 1;
    size_t seg = tc / L, j = tc % L;
    const u32* base = x + seg * len * W;
    u32 acc[S];
    load_elem<S>(acc, base + j * W);
    size_t cnt = (len - j + L - 1) / L;            // elements j, j+L, ... < len
    // every lane of the workgroup runs the same number of rounds (the LDS column is private, so no
    // barrier is involved; lanes with fewer terms multiply by nothing = skip)
    for (size_t k = 1; k < cnt; ++k) {
        const u32* src = base + (j + k * L) * W;
        if constexpr (MUL) {
            load_elem_to_lds<S>(bl, src);
            mont_mul<S>(acc, acc, bl, nn, n0inv);
        } else {
            u32 b[S];
            load_elem<S>(b, src);
            mod_add<S>(acc, acc, b, nmod);
        }
    }
    if constexpr (MUL) canonicalize<S>(acc, nmod);
    if (live) store_elem<S>(out + t * W, acc);
}

// ---------------------------------------------------------------------------------------------
// K8 element-wise ring kernels:  op 0: x + y   op 1: -x   op 2: x*v + y (v one element)   op 3: x*v
// ---------------------------------------------------------------------------------------------
template <int S>
__global__ void __launch_bounds__(BLOCK, (S <= 74 ? 2 : 1))
k_ring_elementwise(u32* __restrict__ out, const u32* __restrict__ x, const u32* __restrict__ y, const u32* __restrict__ v,
                   int op, size_t n, const u32* __restrict__ nmod, u32 n0inv) {
    constexpr int W = stride_for_limbs(S);
    extern __shared__ u32 lds[];
    u32* bl = lds + threadIdx.x;
    size_t el = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    bool live = el < n;
    size_t ec = live ? el : n - 1;
    u32 a[S], r[S];
    load_elem<S>(a, x + ec * W);
    if (op == 0) {
        u32 b[S];
        load_elem<S>(b, y + ec * W);
        mod_add<S>(r, a, b, nmod);
    } else if (op == 1) {
        // N - a, and 0 stays 0
        int32_t borrow = 0;
        u32 nz = 0;
#pragma unroll
        for (int j = 0; j < S; ++j) {
            int32_t d = (int32_t)nmod[j] - (int32_t)a[j] + borrow;
            r[j] = (u32)d & LIMB_MASK;
            borrow = d >> LIMB_BITS;
            nz |= a[j];
        }
#pragma unroll
        for (int j = 0; j < S; ++j) r[j] = nz ? r[j] : 0u;
    } else {
        u32 nn[S];
        load_modulus<S>(nn, nmod);
        load_elem_to_lds<S>(bl, v);
        u32 t[S], b[S];
        mont_mul<S>(t, a, bl, nn, n0inv);
        canonicalize<S>(t, nmod);
        if (op == 2) {
            load_elem<S>(b, y + ec * W);
            mod_add<S>(r, t, b, nmod);
        } else {
#pragma unroll
            for (int j = 0; j < S; ++j) r[j] = t[j];
        }
    }
    if (live) store_elem<S>(out + el * W, r);
}

// ---------------------------------------------------------------------------------------------
// K8 scans.  An affine recurrence x[i] = x[i-1]*e[i] + b[i] (recLin; b == nullptr: prods,
// y[i] = y[i-1]*e[i]) over segments of `seglen` elements (seglen % C == 0 or one segment),
// processed in chunks of C consecutive elements per lane:
//   k_scan_totals : per chunk, the composed map (E = prod e, X = value reached from 0)
//   (recursion on the totals gives every chunk's incoming value)
//   k_scan_apply  : per chunk, replay the recurrence from the incoming value and store x[i]
// `rev`: element i of a segment is read/written at position seglen-1-i (suffix scans of K3).
// ---------------------------------------------------------------------------------------------
// WANT_X = false: Etot[c] = prod of the chunk's e.   WANT_X = true: Xtot[c] = value reached from 0.
// (Two launches for recLin: one accumulator per kernel keeps a + columns + accumulator within 256 VGPRs.)
template <int S, bool WANT_X>
__global__ void __launch_bounds__(BLOCK, (S <= 74 ? 2 : 1))
k_scan_totals(u32* __restrict__ tot, const u32* __restrict__ e, const u32* __restrict__ b,
              size_t n, size_t C, size_t seglen, int rev, const u32* __restrict__ nmod, u32 n0inv,
              const u32* __restrict__ one_m) {
    constexpr int W = stride_for_limbs(S);
    extern __shared__ u32 lds[];
    u32* bl = lds + threadIdx.x;
    u32 nn[S];
    load_modulus<S>(nn, nmod);
    size_t nchunks = (n + C - 1) / C;
    size_t c = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    bool live = c < nchunks;
    size_t cc = live ? c : nchunks - 1;
    size_t lo = cc * C, hi = lo + C < n ? lo + C : n;
    u32 A[S];
#pragma unroll
    for (int j = 0; j < S; ++j) A[j] = WANT_X ? 0u : one_m[j];
    for (size_t i = lo; i < lo + C; ++i) {          // uniform trip count; short chunks idle at the end
        if (i < hi) {
            size_t pos = rev ? (i / seglen) * seglen + (seglen - 1 - i % seglen) : i;
            load_elem_to_lds<S>(bl, e + pos * W);
            mont_mul<S>(A, A, bl, nn, n0inv);
            if constexpr (WANT_X) {
                canonicalize<S>(A, nmod);
                u32 bb[S];
                load_elem<S>(bb, b + pos * W);
                mod_add<S>(A, A, bb, nmod);
            }
        }
    }
    canonicalize<S>(A, nmod);
    if (live) store_elem<S>(tot + c * W, A);
}

// incoming: per-chunk inclusive results of the level above (chunk c starts from incoming[c-1]),
// nullptr = every chunk starts fresh.  A chunk that begins a segment starts fresh (0 / one).
template <int S>
__global__ void __launch_bounds__(BLOCK, (S <= 74 ? 2 : 1))
k_scan_apply(u32* __restrict__ out, const u32* __restrict__ e, const u32* __restrict__ b, const u32* __restrict__ incoming,
             size_t n, size_t C, size_t seglen, int rev, const u32* __restrict__ nmod, u32 n0inv,
             const u32* __restrict__ one_m) {
    constexpr int W = stride_for_limbs(S);
    extern __shared__ u32 lds[];
    u32* bl = lds + threadIdx.x;
    u32 nn[S];
    load_modulus<S>(nn, nmod);
    size_t nchunks = (n + C - 1) / C;
    size_t c = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    bool live = c < nchunks;
    size_t cc = live ? c : nchunks - 1;
    size_t lo = cc * C, hi = lo + C < n ? lo + C : n;
    bool fresh = incoming == nullptr || (lo % seglen) == 0;
    u32 X[S];
    if (fresh) {
#pragma unroll
        for (int j = 0; j < S; ++j) X[j] = b ? 0u : one_m[j];
    } else {
        load_elem<S>(X, incoming + (cc - 1) * W);
    }
    for (size_t i = lo; i < lo + C; ++i) {
        if (i < hi) {
            size_t pos = rev ? (i / seglen) * seglen + (seglen - 1 - i % seglen) : i;
            load_elem_to_lds<S>(bl, e + pos * W);
            u32 t[S];
            mont_mul<S>(t, X, bl, nn, n0inv);
            canonicalize<S>(t, nmod);
            if (b) {
                u32 bb[S];
                load_elem<S>(bb, b + pos * W);
                mod_add<S>(X, t, bb, nmod);
            } else {
#pragma unroll
                for (int j = 0; j < S; ++j) X[j] = t[j];
            }
            if (live) store_elem<S>(out + pos * W, X);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K2 fixed base.  Table T[k][d] = base^(d * 2^(w*k)), k < nwin, d < 2^w, rows of W words at
// (k*2^w + d)*W.  The host supplies sq[j] = base^(2^j) (the sequential squaring chain); level l
// fills d in (2^l, 2^(l+1)):  T[k][d] = T[k][d - 2^l] * T[k][2^l].
// ---------------------------------------------------------------------------------------------
template <int S>
__global__ void __launch_bounds__(BLOCK) k_fixed_seed(u32* __restrict__ T, const u32* __restrict__ sq, int w, int nwin,
                                                      const u32* __restrict__ one_m) {
    constexpr int W = stride_for_limbs(S);
    // one thread per (k, l) plus the d = 0 rows
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    size_t total = (size_t)nwin * (w + 1);
    if (t >= total) return;
    int k = (int)(t / (w + 1)), l = (int)(t % (w + 1));
    u32* dst;
    const u32* src;
    if (l == w) {                          // d = 0: the Montgomery one
        dst = T + ((size_t)k << w) * W;
        src = one_m;
    } else {
        dst = T + (((size_t)k << w) + ((size_t)1 << l)) * W;
        src = sq + ((size_t)k * w + l) * W;
    }
    for (int j = 0; j < W; ++j) dst[j] = j < S ? src[j] : 0u;
}

template <int S>
__global__ void __launch_bounds__(BLOCK, (S <= 74 ? 2 : 1))
k_fixed_level(u32* __restrict__ T, int w, int nwin, int l, const u32* __restrict__ nmod, u32 n0inv) {
    constexpr int W = stride_for_limbs(S);
    extern __shared__ u32 lds[];
    u32* bl = lds + threadIdx.x;
    u32 nn[S];
    load_modulus<S>(nn, nmod);
    size_t per = ((size_t)1 << l) - 1;                 // d = 2^l + 1 .. 2^(l+1) - 1
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    bool live = t < per * nwin;
    size_t tc = live ? t : per * nwin - 1;
    size_t k = tc / per, r = tc % per + 1;             // r = d - 2^l in [1, 2^l)
    u32* row = T + (k << w) * W;
    u32 a[S];
    load_elem<S>(a, row + r * W);
    load_elem_to_lds<S>(bl, row + ((size_t)1 << l) * W);
    u32 o[S];
    mont_mul<S>(o, a, bl, nn, n0inv);
    canonicalize<S>(o, nmod);
    if (live) store_elem<S>(row + (((size_t)1 << l) + r) * W, o);
}

// out[i] = prod_k T[k][digit_k(e[i])]
template <int S>
__global__ void __launch_bounds__(BLOCK, (S <= 74 ? 2 : 1))
k_fixed_exp(u32* __restrict__ out, const u32* __restrict__ T, int w, int nwin, const u32* __restrict__ e, int ewords,
            size_t n, const u32* __restrict__ nmod, u32 n0inv) {
    constexpr int W = stride_for_limbs(S);
    extern __shared__ u32 lds[];
    u32* bl = lds + threadIdx.x;
    u32 nn[S];
    load_modulus<S>(nn, nmod);
    const size_t ntiles = (n + BLOCK - 1) / BLOCK;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        size_t el = t * BLOCK + threadIdx.x;
        bool live = el < n;
        size_t ec = live ? el : n - 1;
        const u32* ep = e + ec * ewords;
        u32 a[S];
        u32 d = exp_digit(ep, ewords, 0, w);
        load_elem<S>(a, T + (size_t)d * W);
#pragma unroll 1
        for (int k = 1; k < nwin; ++k) {
            d = exp_digit(ep, ewords, k * w, w);
            load_elem_to_lds<S>(bl, T + (((size_t)k << w) + d) * W);
            mont_mul<S>(a, a, bl, nn, n0inv);
        }
        canonicalize<S>(a, nmod);
        if (live) store_elem<S>(out + el * W, a);
    }
}

// ---------------------------------------------------------------------------------------------
// K3 multi-exponentiation (Pippenger): counting sort of (window, digit) then one lane per bucket.
// ---------------------------------------------------------------------------------------------
// counts[win][d] += 1 for every element; one thread per (element, window).
__global__ void __launch_bounds__(BLOCK) k_bucket_hist(u32* __restrict__ counts, const u32* __restrict__ e, int ewords,
                                                       size_t n, int c, int nwin) {
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    size_t total = n * (size_t)nwin;
    for (; t < total; t += (size_t)gridDim.x * BLOCK) {
        size_t i = t % n;
        int w = (int)(t / n);
        u32 d = exp_digit(e + i * ewords, ewords, w * c, c);
        atomicAdd(&counts[((size_t)w << c) + d], 1u);
    }
}
// ---- generic exclusive scan of a u32 array (three tiny kernels; n up to a few million) ----------
constexpr int SCAN_ITEMS = 16;                         // items per thread
__global__ void __launch_bounds__(BLOCK) k_u32_blocksum(u32* __restrict__ bsum, const u32* __restrict__ in, size_t n) {
    __shared__ u32 part[BLOCK];
    size_t base = ((size_t)blockIdx.x * BLOCK + threadIdx.x) * SCAN_ITEMS;
    u32 s = 0;
    for (int k = 0; k < SCAN_ITEMS; ++k) s += base + k < n ? in[base + k] : 0u;
    part[threadIdx.x] = s;
    __syncthreads();
    for (int st = BLOCK / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) part[threadIdx.x] += part[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) bsum[blockIdx.x] = part[0];
}
__global__ void k_u32_scan_top(u32* __restrict__ bsum, size_t nblocks, u32* __restrict__ total) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        u32 run = 0;
        for (size_t i = 0; i < nblocks; ++i) {
            u32 v = bsum[i];
            bsum[i] = run;
            run += v;
        }
        *total = run;
    }
}
// out[i] = exclusive prefix; out2 (optional) receives a copy (the scatter cursors); out[n] = total
__global__ void __launch_bounds__(BLOCK) k_u32_scan_apply(u32* __restrict__ out, u32* __restrict__ out2,
                                                          const u32* __restrict__ in, const u32* __restrict__ bsum, size_t n) {
    __shared__ u32 part[BLOCK];
    size_t base = ((size_t)blockIdx.x * BLOCK + threadIdx.x) * SCAN_ITEMS;
    u32 v[SCAN_ITEMS];
    u32 s = 0;
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = base + k < n ? in[base + k] : 0u;
        s += v[k];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 run = bsum[blockIdx.x];
        for (int i = 0; i < BLOCK; ++i) {
            u32 t = part[i];
            part[i] = run;
            run += t;
        }
    }
    __syncthreads();
    u32 run = part[threadIdx.x];
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k < n) {
            out[base + k] = run;
            if (out2) out2[base + k] = run;
        }
        run += v[k];
        if (base + k + 1 == n) out[n] = run;
    }
}

// sorted[cursor[bucket]++] = element index; one thread per (element, window); bucket = w*2^c + digit
__global__ void __launch_bounds__(BLOCK) k_bucket_scatter(u32* __restrict__ sorted, u32* __restrict__ cursor,
                                                          const u32* __restrict__ e, int ewords, size_t n, int c, int nwin) {
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    size_t total = n * (size_t)nwin;
    for (; t < total; t += (size_t)gridDim.x * BLOCK) {
        size_t i = t % n;
        int w = (int)(t / n);
        u32 d = exp_digit(e + i * ewords, ewords, w * c, c);
        u32 pos = atomicAdd(&cursor[((size_t)w << c) + d], 1u);
        sorted[pos] = (u32)i;
    }
}
// digit 0 contributes nothing: drop those buckets' items
__global__ void __launch_bounds__(BLOCK) k_bucket_drop_zero(u32* __restrict__ counts, int c, int nwin) {
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t < (size_t)nwin) counts[t << c] = 0;
}
// cnt_out[b] = ceil(cnt_in[b] / F); maxcnt = max over b
__global__ void __launch_bounds__(BLOCK) k_task_counts(u32* __restrict__ cnt_out, const u32* __restrict__ cnt_in,
                                                       size_t nbuckets, u32 F, u32* __restrict__ maxcnt) {
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    u32 v = 0;
    if (t < nbuckets) {
        v = (cnt_in[t] + F - 1) / F;
        cnt_out[t] = v;
    }
    for (int o = 32; o > 0; o >>= 1) v = max(v, (u32)__shfl_xor((int)v, o));
    if ((threadIdx.x & 63) == 0 && v) atomicMax(maxcnt, v);
}

// One level of the per-bucket product tree.  Bucket b owns cnt_in[b] items at off_in[b]; output item
// (b, j) = product of its input items [jF, (j+1)F) and lands at off_out[b] + j.  One lane per output
// item, so a bucket of any size is spread over ceil(size/F) lanes: no lane ever walks a long bucket
// (skewed digits -- a short top window, equal exponents -- would otherwise serialise on one lane).
// FIRST: input items are rows of x selected through `sorted`; otherwise rows of `in`.
template <int S, bool FIRST>
__global__ void __launch_bounds__(BLOCK, (S <= 74 ? 2 : 1))
k_bucket_level(u32* __restrict__ out, const u32* __restrict__ in, const u32* __restrict__ sorted,
               const u32* __restrict__ off_in, const u32* __restrict__ cnt_in, const u32* __restrict__ off_out,
               size_t nbuckets, size_t total_out, u32 F, const u32* __restrict__ nmod, u32 n0inv) {
    constexpr int W = stride_for_limbs(S);
    extern __shared__ u32 lds[];
    u32* bl = lds + threadIdx.x;
    u32 nn[S];
    load_modulus<S>(nn, nmod);
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    bool live = t < total_out;
    size_t tc = live ? t : total_out - 1;
    // b = last bucket with off_out[b] <= tc  (empty buckets share their successor's offset)
    size_t lo = 0, hi = nbuckets;                 // invariant: off_out[lo] <= tc < off_out[hi] (off_out[nbuckets] = total)
    while (hi - lo > 1) {
        size_t mid = (lo + hi) >> 1;
        if (off_out[mid] <= tc) lo = mid; else hi = mid;
    }
    size_t b = lo;
    u32 j = (u32)(tc - off_out[b]);
    u32 start = off_in[b] + j * F;
    u32 end = off_in[b] + cnt_in[b];
    if (end > start + F) end = start + F;
    u32 acc[S];
    auto row = [&](u32 k) -> const u32* { return FIRST ? in + (size_t)sorted[k] * W : in + (size_t)k * W; };
    load_elem<S>(acc, row(start));
    for (u32 k = start + 1; k < end; ++k) {
        load_elem_to_lds<S>(bl, row(k));
        mont_mul<S>(acc, acc, bl, nn, n0inv);
    }
    canonicalize<S>(acc, nmod);
    if (live) store_elem<S>(out + t * W, acc);
}
// B[b] = the bucket's single remaining item, or one if it is empty
template <int S>
__global__ void __launch_bounds__(BLOCK) k_bucket_finalize(uint4* __restrict__ B, const uint4* __restrict__ items,
                                                           const u32* __restrict__ off_in, const u32* __restrict__ cnt_in,
                                                           size_t nbuckets, const uint4* __restrict__ one_row) {
    constexpr int CPR = stride_for_limbs(S) / 4;
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    size_t total = nbuckets * CPR;
    for (; t < total; t += (size_t)gridDim.x * BLOCK) {
        size_t b = t / CPR;
        int ch = (int)(t % CPR);
        B[t] = cnt_in[b] ? items[(size_t)off_in[b] * CPR + ch] : one_row[ch];
    }
}
// overwrite element 0 of every segment with `one` (the d = 0 slot of the suffix products)
template <int S>
__global__ void __launch_bounds__(BLOCK) k_set_segment_heads(u32* __restrict__ a, size_t seglen, size_t nseg,
                                                             const u32* __restrict__ one_m) {
    constexpr int W = stride_for_limbs(S);
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= nseg) return;
    u32* dst = a + t * seglen * W;
    for (int j = 0; j < W; ++j) dst[j] = j < S ? one_m[j] : 0u;
}

}  // namespace vmn
