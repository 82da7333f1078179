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
                mont_mul<S>(a, a, bl, nn, n0inv);
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
