// modp_kernels.h — gfx950 kernels over arrays of residues in M28 form (see mont28.h).
//
// Device array layout ("AoS"): element i occupies W consecutive 32-bit words at base + i*W, limb j
// (28 bits) in one word, padding words zero.  Elements are contiguous so that gathers / permutations /
// sharding by contiguous ranges move whole 16-byte-aligned rows.
//
// Two execution geometries, chosen by the modulus size (struct Cfg):
//   LPE = 1  one element per lane (moduli up to 2072 bits, S <= 74 limbs): the lane's multiplicand sits
//            in VGPRs, its multiplier is staged in LDS as lds[limb*256 + tid], the modulus limbs are
//            wave-uniform SGPRs.
//   LPE = 2  two lanes per element (3072-bit moduli, S = 110): a lane holds L = S/2 limbs and columns
//            (a + columns + modulus half = 4L = 220 VGPRs; one lane would need 330 > the 256
//            architectural VGPRs).  The lanes of a pair exchange the reduction factor and one column
//            per row through DPP (mont28.h / the generator).  In memory a lane's share is LW = 56
//            words (16-byte aligned), W = 2*LW.
//   LPE = 4  four lanes per element (4096-bit moduli, S = 148, L = 37: two lanes would need 4*74 = 296
//            VGPRs).  Same scheme inside a quad of lanes (quad_perm DPP); carries / borrows cross the
//            three lane boundaries by repeated sweeps.
//   LPE = 8 / 16  moduli up to 8192 / 16384 bits (S = 296 / 592, L = 37): half a DPP row resp. a whole one (row shifts
//            with bank masks); the same kernels, the columns relieved every 74 rows.  Built for completeness -- the
//            reference offers safe primes up to 15 424 bits and benchmarks with a 15 492-bit group -- not tuned.
#pragma once
#include "mont28.h"

namespace vmn {

constexpr int BLOCK = 256;   // threads per workgroup: 4 waves, one per SIMD

// "Wide" geometries for SMALL arrays (fewer elements than the chip has lanes at one element per lane): the same
// element rows -- same R, same words in memory -- are worked on by four lanes, so that a chain of dependent products is
// ~2.5 times shorter and four times as many waves are in flight.
//   Cfg<76, 4>   the wide form of Cfg<74, 1> (2048-bit moduli): four shares of 19 columns = 76 columns, the two above the 74
//                limbs are zero; a product still takes ROWS = 74 reduction rows, so R = 2^(28*74).
//   Cfg<112, 4>  the wide form of Cfg<110, 2> (3072-bit moduli): four shares of 28 columns, ROWS = 110.
// A wide geometry reads and writes the rows of its BASE geometry: limb g of the element is word
// (g / BASE_L) * BASE_LW + g % BASE_L of the row (BASE_L limbs per share of the base layout, BASE_LW words apart), the
// columns g >= ROWS exist in registers only.
//   Cfg<80, 8>   the widest form of Cfg<74, 1>, for the smallest arrays: eight shares of 10 columns, ROWS = 74.
__host__ __device__ constexpr int rows_for(int S, int LPE) {
    return ((S == 76 && LPE == 4) || (S == 80 && LPE == 8)) ? 74 : (S == 112 && LPE == 4) ? 110 : S;
}
__host__ __device__ constexpr int base_lpe_for(int S, int LPE) {
    return ((S == 76 && LPE == 4) || (S == 80 && LPE == 8)) ? 1 : (S == 112 && LPE == 4) ? 2 : LPE;
}

template <int S_, int LPE_>
struct Cfg {
    static constexpr int S = S_;                        // columns per element (limbs, plus the zero columns of a wide geometry)
    static constexpr int LPE = LPE_;                    // lanes per element
    static constexpr int L = S_ / LPE_;                 // limbs per lane
    static constexpr int ROWS = rows_for(S_, LPE_);     // reduction rows of a product: R = 2^(28 ROWS)
    static constexpr int BASE_LPE = base_lpe_for(S_, LPE_);
    static constexpr bool WIDE = BASE_LPE != LPE_;      // works on the rows of another geometry's layout
    static constexpr int BASE_L = ROWS / BASE_LPE;      // limbs per share of the memory layout
    static constexpr int BASE_LW = stride_for_limbs(BASE_L);
    static constexpr int LW = WIDE ? 0 : BASE_LW;       // words of one lane's share in memory (own layout only)
    static constexpr int W = BASE_LPE * BASE_LW;        // words per element in memory
    static constexpr int EPB = BLOCK / LPE_;            // elements per workgroup
    static constexpr int MINW = 2;                      // waves per SIMD the kernels are built for
    static_assert(S_ % LPE_ == 0 && ROWS % BASE_LPE == 0, "limbs must split evenly over the lanes of an element");
    // word of limb g (< ROWS) in the row
    static __host__ __device__ constexpr int word_of(int g) { return (g / BASE_L) * BASE_LW + g % BASE_L; }
};

// What a lane needs to know about its place: element slot in the workgroup, which half it holds, its
// LDS column (limb i of the element's multiplier is bl[i*EPB]).
template <class C>
struct Lane {
    int eslot;        // element slot within the workgroup
    int half;         // 0 .. LPE-1: which share of the element this lane holds
    u32 lowmask;      // LPE > 1: 0xffffffff on lane 0 of the element, 0 elsewhere
    u32 nottopmask;   // LPE > 1: 0xffffffff on every lane of the element but the last
    u32* bl;
    __device__ __forceinline__ explicit Lane(u32* lds) {
        eslot = threadIdx.x / C::LPE;
        half = threadIdx.x % C::LPE;
        lowmask = half == 0 ? 0xffffffffu : 0u;
        nottopmask = half == C::LPE - 1 ? 0u : 0xffffffffu;
        bl = lds + eslot;
    }
};

// Cross-lane moves inside an element (DPP; elements are aligned groups of 2, 4 or 8 lanes: quad_perm within a quad, row
// shifts with bank masks for the two quads of an 8-lane element -- bank k of a DPP row = its lanes 4k .. 4k+3).
// from_below: the value of lane h-1 (lane 0 gets its own / a neighbour's: mask with ~lowmask); from_top: the last lane's
// value on all lanes; or_all: OR over the element's lanes, on all lanes.
template <int LPE>
__device__ __forceinline__ u32 from_below(u32 x) { return lane_below<LPE>(x); }
// lower quads <- upper quads (row_shl:4 into banks 0, 2) resp. upper <- lower (row_shr:4 into banks 1, 3); other lanes keep old
__device__ __forceinline__ u32 quad_from_upper(u32 old, u32 x) { return (u32)__builtin_amdgcn_update_dpp((int)old, (int)x, 0x104, 0xf, 0x5, false); }
__device__ __forceinline__ u32 quad_from_lower(u32 old, u32 x) { return (u32)__builtin_amdgcn_update_dpp((int)old, (int)x, 0x114, 0xf, 0xA, false); }
// (sixteen lanes per element = a whole DPP row: one more step of each kind)
template <int LPE>
__device__ __forceinline__ u32 from_top(u32 x) {
    if constexpr (LPE == 2) return (u32)__builtin_amdgcn_mov_dpp((int)x, 0xF5, 0xf, 0xf, true);     // [1,1,3,3]
    else {
        u32 t = (u32)__builtin_amdgcn_mov_dpp((int)x, 0xFF, 0xf, 0xf, true);                        // [3,3,3,3]
        if constexpr (LPE == 8) t = quad_from_upper(t, t);
        if constexpr (LPE == 16) {
            t = (u32)__builtin_amdgcn_update_dpp((int)t, (int)t, 0x104, 0xf, 0x4, false);           // lanes 8-11 <- 12-15 (row_shl:4, bank 2)
            t = (u32)__builtin_amdgcn_update_dpp((int)t, (int)t, 0x108, 0xf, 0x3, false);           // lanes 0-7 <- 8-15 (row_shl:8, banks 0, 1)
        }
        return t;
    }
}
// the value of lane h+1 of the element (the last lane receives its own / a neighbour's: mask with nottopmask); lane 0's value on all lanes
template <int LPE>
__device__ __forceinline__ u32 from_above(u32 x) {
    if constexpr (LPE == 2) return (u32)__builtin_amdgcn_mov_dpp((int)x, 0xF5, 0xf, 0xf, true);     // [1,1,3,3]
    else if constexpr (LPE == 4) return (u32)__builtin_amdgcn_mov_dpp((int)x, 0xF9, 0xf, 0xf, true);   // [1,2,3,3]
    else return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x101, 0xf, 0xf, true);                 // row_shl:1 (8 or 16 lanes)
}
template <int LPE>
__device__ __forceinline__ u32 from_lane0(u32 x) {
    if constexpr (LPE == 2) return (u32)__builtin_amdgcn_mov_dpp((int)x, 0xA0, 0xf, 0xf, true);     // [0,0,2,2]
    else {
        u32 t = (u32)__builtin_amdgcn_mov_dpp((int)x, 0x00, 0xf, 0xf, true);                        // [0,0,0,0]
        if constexpr (LPE == 8) t = quad_from_lower(t, t);
        if constexpr (LPE == 16) {
            t = (u32)__builtin_amdgcn_update_dpp((int)t, (int)t, 0x114, 0xf, 0x2, false);           // lanes 4-7 <- 0-3 (row_shr:4, bank 1)
            t = (u32)__builtin_amdgcn_update_dpp((int)t, (int)t, 0x118, 0xf, 0xC, false);           // lanes 8-15 <- 0-7 (row_shr:8, banks 2, 3)
        }
        return t;
    }
}
template <int LPE>
__device__ __forceinline__ u32 or_all(u32 x) {
    x |= (u32)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xf, 0xf, true);                               // [1,0,3,2]
    if constexpr (LPE >= 4) x |= (u32)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xf, 0xf, true);       // [2,3,0,1]
    if constexpr (LPE == 8) x |= quad_from_upper(0u, x) | quad_from_lower(0u, x);
    if constexpr (LPE == 16) {
        x |= (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x124, 0xf, 0xf, false);                   // row_ror:4
        x |= (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xf, 0xf, false);                   // row_ror:8
    }
    return x;
}

// ---------------------------------------------------------------------------------------------
// element movement helpers
// ---------------------------------------------------------------------------------------------
template <class C>
__device__ __forceinline__ void load_elem(u32 (&a)[C::L], const u32* __restrict__ p, const Lane<C>& ln) {
    if constexpr (C::WIDE) {                            // shares cut across the base layout: dword loads by limb index
#pragma unroll
        for (int j = 0; j < C::L; ++j) {
            const int g = ln.half * C::L + j;
            a[j] = g < C::ROWS ? p[C::word_of(g)] : 0u;
        }
        return;
    }
    const uint4* q = reinterpret_cast<const uint4*>(p + ln.half * C::LW);
#pragma unroll
    for (int k = 0; k < C::LW / 4; ++k) {
        uint4 v = q[k];
        if (4 * k + 0 < C::L) a[4 * k + 0] = v.x;
        if (4 * k + 1 < C::L) a[4 * k + 1] = v.y;
        if (4 * k + 2 < C::L) a[4 * k + 2] = v.z;
        if (4 * k + 3 < C::L) a[4 * k + 3] = v.w;
    }
}
template <class C>
__device__ __forceinline__ void store_elem(u32* __restrict__ p, const u32 (&a)[C::L], const Lane<C>& ln) {
    if constexpr (C::WIDE) {                            // (the columns above the limbs are zero: value < 2N < 2^(28 ROWS))
#pragma unroll
        for (int j = 0; j < C::L; ++j) {
            const int g = ln.half * C::L + j;
            if (g < C::ROWS) p[C::word_of(g)] = a[j];
        }
        if (ln.half < C::BASE_LPE) {                    // the padding words of the base layout stay zero
#pragma unroll
            for (int k = C::BASE_L; k < C::BASE_LW; ++k) p[ln.half * C::BASE_LW + k] = 0u;
        }
        return;
    }
    uint4* q = reinterpret_cast<uint4*>(p + ln.half * C::LW);
#pragma unroll
    for (int k = 0; k < C::LW / 4; ++k) {
        uint4 v;
        v.x = 4 * k + 0 < C::L ? a[4 * k + 0] : 0;
        v.y = 4 * k + 1 < C::L ? a[4 * k + 1] : 0;
        v.z = 4 * k + 2 < C::L ? a[4 * k + 2] : 0;
        v.w = 4 * k + 3 < C::L ? a[4 * k + 3] : 0;
        q[k] = v;
    }
}
// global element -> the element's LDS column (multiplier operand); each lane moves its own share
template <class C>
__device__ __forceinline__ void load_elem_to_lds(const Lane<C>& ln, const u32* __restrict__ p) {
    u32* dst = ln.bl + ln.half * C::L * C::EPB;
    if constexpr (C::WIDE) {
#pragma unroll
        for (int j = 0; j < C::L; ++j) {
            const int g = ln.half * C::L + j;
            dst[j * C::EPB] = g < C::ROWS ? p[C::word_of(g)] : 0u;
        }
        return;
    }
    const uint4* q = reinterpret_cast<const uint4*>(p + ln.half * C::LW);
#pragma unroll
    for (int k = 0; k < C::LW / 4; ++k) {
        uint4 v = q[k];
        if (4 * k + 0 < C::L) dst[(4 * k + 0) * C::EPB] = v.x;
        if (4 * k + 1 < C::L) dst[(4 * k + 1) * C::EPB] = v.y;
        if (4 * k + 2 < C::L) dst[(4 * k + 2) * C::EPB] = v.z;
        if (4 * k + 3 < C::L) dst[(4 * k + 3) * C::EPB] = v.w;
    }
}
template <class C>
__device__ __forceinline__ void regs_to_lds(const Lane<C>& ln, const u32 (&a)[C::L]) {
    u32* dst = ln.bl + ln.half * C::L * C::EPB;
#pragma unroll
    for (int j = 0; j < C::L; ++j) dst[j * C::EPB] = a[j];
}
// a small constant element (value v < 2^28) into the LDS column: limb 0 = v, the rest 0
template <class C>
__device__ __forceinline__ void small_to_lds(const Lane<C>& ln, u32 v) {
    u32* dst = ln.bl + ln.half * C::L * C::EPB;
#pragma unroll
    for (int j = 0; j < C::L; ++j) dst[j * C::EPB] = (j == 0 && ln.half == 0) ? v : 0u;
}
// device constant (modulus-like layout) -> LDS column
template <class C>
__device__ __forceinline__ void const_to_lds(const Lane<C>& ln, const u32* __restrict__ c) {
    u32* dst = ln.bl + ln.half * C::L * C::EPB;
#pragma unroll
    for (int j = 0; j < C::L; ++j) {
        if constexpr (C::WIDE) {
            const int g = ln.half * C::L + j;
            dst[j * C::EPB] = g < C::ROWS ? c[C::word_of(g)] : 0u;
        } else {
            dst[j * C::EPB] = c[ln.half * C::LW + j];
        }
    }
}
// this lane's share of a constant in modulus layout.  LPE = 1: uniform address => scalar loads into SGPRs.
template <class C>
__device__ __forceinline__ void load_modulus(u32 (&n)[C::L], const u32* __restrict__ nmod, const Lane<C>& ln) {
#pragma unroll
    for (int j = 0; j < C::L; ++j) {
        if constexpr (C::WIDE) {
            const int g = ln.half * C::L + j;
            n[j] = g < C::ROWS ? nmod[C::word_of(g)] : 0u;
        } else {
            n[j] = C::LPE == 1 ? nmod[j] : nmod[ln.half * C::LW + j];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// carry / borrow chains (within a lane, and across the two lanes of a pair)
// ---------------------------------------------------------------------------------------------
// out = limbs of sum_j v[j] 2^(28j) (+ carry-in), v[j] arbitrary 64-bit columns; returns the carry out
template <int L>
__device__ __forceinline__ u64 carry_sweep(u32 (&out)[L], const u64 (&v)[L], u64 cin) {
    u64 c = cin;
#pragma unroll
    for (int j = 0; j < L; ++j) {
        c += v[j];
        out[j] = (u32)c & LIMB_MASK;
        c >>= LIMB_BITS;
    }
    return c;
}
// Resolve lazy columns into 28-bit limbs.  Several lanes per element: a lane's carry-out enters column 0 of the
// lane above; after sweep t the lanes 0..t are final (lane 0 repeats its sweep with carry 0 = same result), so
// LPE sweeps settle the element.
template <class C>
__device__ __forceinline__ void normalize(u32 (&out)[C::L], const u64 (&T)[C::L], const Lane<C>& ln) {
    u64 c = carry_sweep<C::L>(out, T, 0);
    if constexpr (C::LPE > 1) {
#pragma unroll
        for (int t = 1; t < C::LPE; ++t) {
            u32 lo = from_below<C::LPE>((u32)c) & ~ln.lowmask, hi = from_below<C::LPE>((u32)(c >> 32)) & ~ln.lowmask;
            c = carry_sweep<C::L>(out, T, ((u64)hi << 32) | lo);
        }
    }
}
// d = x - n (limbs, borrow-in bin = 0 / -1); returns borrow-out (0 / -1)
template <int L>
__device__ __forceinline__ int32_t borrow_sweep(u32 (&d)[L], const u32 (&x)[L], const u32 (&n)[L], int32_t bin) {
    int32_t borrow = bin;
#pragma unroll
    for (int j = 0; j < L; ++j) {
        int32_t v = (int32_t)x[j] - (int32_t)n[j] + borrow;      // limbs < 2^28: no int32 overflow
        d[j] = (u32)v & LIMB_MASK;
        borrow = v >> LIMB_BITS;
    }
    return borrow;
}
// element-wide x - n: d and the final borrow (0: x >= n, -1: x < n), identical on all lanes of the element
template <class C>
__device__ __forceinline__ int32_t sub_full(u32 (&d)[C::L], const u32 (&x)[C::L], const u32 (&n)[C::L], const Lane<C>& ln) {
    int32_t b = borrow_sweep<C::L>(d, x, n, 0);
    if constexpr (C::LPE > 1) {
#pragma unroll
        for (int t = 1; t < C::LPE; ++t) {
            int32_t bin = (int32_t)(from_below<C::LPE>((u32)b) & ~ln.lowmask);
            b = borrow_sweep<C::L>(d, x, n, bin);
        }
        b = (int32_t)from_top<C::LPE>((u32)b);
    }
    return b;
}
// x (limbs, value < 2N) -> canonical x mod N (< N), branch-free
template <class C>
__device__ __forceinline__ void canonicalize(u32 (&x)[C::L], const u32 (&n)[C::L], const Lane<C>& ln) {
    u32 d[C::L];
    bool ge = sub_full<C>(d, x, n, ln) == 0;
#pragma unroll
    for (int j = 0; j < C::L; ++j) x[j] = ge ? d[j] : x[j];
}
// r = a + b mod N (a, b canonical).  32-bit sums (limbs < 2^28, so a[j] + b[j] + carry < 2^30): the 64-bit
// column sweep would double the live registers, which the pair geometry (modulus half in VGPRs) cannot afford.
template <class C>
__device__ __forceinline__ void mod_add(u32 (&r)[C::L], const u32 (&a)[C::L], const u32 (&b)[C::L], const u32 (&n)[C::L],
                                        const Lane<C>& ln) {
    // r may alias a or b: sums first (own registers), then the carry sweep(s), then r is written
    u32 s[C::L];
#pragma unroll
    for (int j = 0; j < C::L; ++j) s[j] = a[j] + b[j];
    u32 cin = 0;
    if constexpr (C::LPE > 1) {
#pragma unroll
        for (int t = 1; t < C::LPE; ++t) {              // carry into this lane, settled from lane 0 upwards
            u32 c = cin;
#pragma unroll
            for (int j = 0; j < C::L; ++j) c = (c + s[j]) >> LIMB_BITS;
            cin = from_below<C::LPE>(c) & ~ln.lowmask;
        }
    }
    u32 c = cin;
#pragma unroll
    for (int j = 0; j < C::L; ++j) {
        c += s[j];
        r[j] = c & LIMB_MASK;
        c >>= LIMB_BITS;
    }
    canonicalize<C>(r, n, ln);
}
// r = -a mod N
template <class C>
__device__ __forceinline__ void mod_neg(u32 (&r)[C::L], const u32 (&a)[C::L], const u32 (&n)[C::L], const Lane<C>& ln) {
    u32 d[C::L];
    sub_full<C>(d, n, a, ln);                    // N - a  (a < N)
    u32 nz = 0;
#pragma unroll
    for (int j = 0; j < C::L; ++j) nz |= a[j];
    if constexpr (C::LPE > 1) nz = or_all<C::LPE>(nz);
#pragma unroll
    for (int j = 0; j < C::L; ++j) r[j] = nz ? d[j] : 0u;
}

// r = a * (element's LDS column) / R mod N, limbs normalised, value < 2N
template <class C>
__device__ __forceinline__ void mont_mul(u32 (&r)[C::L], const u32 (&a)[C::L], const Lane<C>& ln, const u32 (&n)[C::L], u32 n0inv) {
    u64 T[C::L];
    if constexpr (C::LPE == 1) {
        mont_mul_columns<C::L>(T, a, ln.bl, C::EPB, n, n0inv);
    } else {
        mont_mul_columns_lanes<C::L, C::LPE, C::ROWS>(T, a, ln.bl, C::EPB, n, n0inv, ln.lowmask, ln.nottopmask);
    }
    normalize<C>(r, T, ln);
}
// r = a^2 / R mod N; the element's LDS column must hold a copy of a
template <class C>
__device__ __forceinline__ void mont_sqr(u32 (&r)[C::L], const u32 (&a)[C::L], const Lane<C>& ln, const u32 (&n)[C::L], u32 n0inv) {
    u64 T[C::L];
    if constexpr (C::LPE == 1) {
        mont_sqr_columns<C::L>(T, a, ln.bl, C::EPB, n, n0inv);
    } else {
        mont_sqr_columns_lanes<C::L, C::LPE, C::ROWS>(T, a, ln.bl, C::EPB, n, n0inv, ln.lowmask, ln.nottopmask);
    }
    normalize<C>(r, T, ln);
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
// word k of the element whose S limbs sit in the LDS column (any lane of the element may ask for any word)
template <class C>
__device__ __forceinline__ u32 word_from_lds(const Lane<C>& ln, int k) {
    int bit = 32 * k, j = bit / 28, sh = bit % 28;
    u64 acc = 0;
    if (j < C::S) acc = (u64)ln.bl[j * C::EPB] >> sh;
    if (j + 1 < C::S) acc |= (u64)ln.bl[(j + 1) * C::EPB] << (28 - sh);
    if (j + 2 < C::S) acc |= (u64)ln.bl[(j + 2) * C::EPB] << (56 - sh);
    return (u32)acc;
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

// this lane's limbs of the big-endian integer at src
template <class C, int NW>
__device__ __forceinline__ void limbs_from_be(u32 (&a)[C::L], const uint8_t* __restrict__ src, long nbytes, const Lane<C>& ln) {
    if constexpr (C::LPE == 1) {
        u32 w[NW];
#pragma unroll
        for (int k = 0; k < NW; ++k) w[k] = load_be_word(src, nbytes, k);
        words_to_limbs<C::L, NW>(a, w);
    } else {
#pragma unroll 1
        for (int j = 0; j < C::L; ++j) {            // the word index depends on the lane's half: read from memory per limb
            int bit = 28 * (ln.half * C::L + j), k = bit >> 5, sh = bit & 31;
            u32 lo = k < NW ? load_be_word(src, nbytes, k) : 0u;
            u32 hi = k + 1 < NW ? load_be_word(src, nbytes, k + 1) : 0u;
            a[j] = (u32)((((u64)hi << 32) | lo) >> sh) & LIMB_MASK;
        }
    }
}

// flags[0] |= 1 if some value >= N ; flags[0] |= 2 if some value == 0 (only reported).
// Out-of-range values are replaced by `one` (x = 1) -- the reference substitutes trivial values
// for malformed input (P/hvzk/PoSBasicTW.java:794-815).
template <class C, int NW>
__global__ void __launch_bounds__(BLOCK, C::MINW)
k_import_be(u32* __restrict__ out, const uint8_t* __restrict__ be, size_t nbytes, size_t stride, int mode, size_t n,
            const u32* __restrict__ nmod, u32 n0inv, const u32* __restrict__ rr, u32* __restrict__ flags) {
    // mode bit 0: every value is preceded by a byte-tree leaf header; bit 1: values >= N are REDUCED mod N instead of
    // being reported and replaced (pseudo-random integers wider than the modulus: a * RR / R is a mod N for every a < R)
    const int leaf_hdr = mode & 1;
    const bool reduce = (mode & 2) != 0;
    extern __shared__ u32 lds[];
    Lane<C> ln(lds);
    u32 nn[C::L];
    load_modulus<C>(nn, nmod, ln);
    size_t el = (size_t)blockIdx.x * C::EPB + ln.eslot;
    bool live = el < n;
    size_t ec = live ? el : n - 1;
    const uint8_t* src = be + ec * stride;             // stride = nbytes, or nbytes + 5 with byte-tree leaf headers
    if (leaf_hdr) {                                   // 01 | uint32_be(nbytes) in front of every value
        bool hdr_ok = src[0] == 1 && src[1] == (uint8_t)(nbytes >> 24) && src[2] == (uint8_t)(nbytes >> 16) &&
                      src[3] == (uint8_t)(nbytes >> 8) && src[4] == (uint8_t)nbytes;
        if (live && !hdr_ok) atomicOr(flags, 4u);
        src += 5;
    }
    u32 extra = 0;                                    // leading bytes beyond NW words must be zero
    for (long o = (long)nbytes - 4L * NW - 1; o >= 0; --o) extra |= src[o];
    u32 a[C::L];
    limbs_from_be<C, NW>(a, src, (long)nbytes, ln);
    u32 d[C::L];
    bool bad = !reduce && (sub_full<C>(d, a, nn, ln) == 0 || extra != 0);     // a >= N
    u32 nz = 0;
#pragma unroll
    for (int j = 0; j < C::L; ++j) nz |= a[j];
    if constexpr (C::LPE > 1) nz = or_all<C::LPE>(nz);
    if (live && bad) atomicOr(flags, 1u);
    if (live && nz == 0) atomicOr(flags, 2u);
    if (bad) {
#pragma unroll
        for (int j = 0; j < C::L; ++j) a[j] = (j == 0 && ln.half == 0) ? 1u : 0u;
    }
    const_to_lds<C>(ln, rr);                          // to Montgomery form: a * RR / R
    u32 r[C::L];
    mont_mul<C>(r, a, ln, nn, n0inv);
    canonicalize<C>(r, nn, ln);
    if (live) store_elem<C>(out + el * C::W, r, ln);
}

// r (canonical standard representative, limbs) -> words; pair mode goes through the LDS column
template <class C, int NW, typename F>
__device__ __forceinline__ void emit_words(const u32 (&r)[C::L], const Lane<C>& ln, F&& put) {
    if constexpr (C::LPE == 1) {
        u32 w[NW];
        limbs_to_words<C::L, NW>(w, r);
#pragma unroll
        for (int k = 0; k < NW; ++k) put(k, w[k]);
    } else {
        regs_to_lds<C>(ln, r);
#pragma unroll 1
        for (int k = ln.half; k < NW; k += C::LPE) put(k, word_from_lds<C>(ln, k));
    }
}

template <class C, int NW>
__global__ void __launch_bounds__(BLOCK, C::MINW)
k_export_be(uint8_t* __restrict__ be, size_t nbytes, size_t stride, int leaf_hdr, const u32* __restrict__ in, size_t n,
            const u32* __restrict__ nmod, u32 n0inv) {
    extern __shared__ u32 lds[];
    Lane<C> ln(lds);
    u32 nn[C::L];
    load_modulus<C>(nn, nmod, ln);
    size_t el = (size_t)blockIdx.x * C::EPB + ln.eslot;
    bool live = el < n;
    size_t ec = live ? el : n - 1;
    u32 a[C::L];
    load_elem<C>(a, in + ec * C::W, ln);
    small_to_lds<C>(ln, 1u);                          // multiply by 1: leaves the Montgomery domain
    u32 r[C::L];
    mont_mul<C>(r, a, ln, nn, n0inv);
    canonicalize<C>(r, nn, ln);
    uint8_t* dst = be + ec * stride;
    if (leaf_hdr) {
        if (live && ln.half == 0) {
            dst[0] = 1;
            dst[1] = (uint8_t)(nbytes >> 24);
            dst[2] = (uint8_t)(nbytes >> 16);
            dst[3] = (uint8_t)(nbytes >> 8);
            dst[4] = (uint8_t)nbytes;
        }
        dst += 5;
    }
    emit_words<C, NW>(r, ln, [&](int k, u32 w) { if (live) store_be_word(dst, (long)nbytes, k, w); });
    if (live && ln.half == 0) {
        for (long o = (long)nbytes - 4L * NW - 1; o >= 0; --o) dst[o] = 0;
    }
}

// M28 form -> packed little-endian words of the standard representative (exponent use)
template <class C, int NW>
__global__ void __launch_bounds__(BLOCK, C::MINW)
k_to_words(u32* __restrict__ out, const u32* __restrict__ in, size_t n, const u32* __restrict__ nmod, u32 n0inv) {
    extern __shared__ u32 lds[];
    Lane<C> ln(lds);
    u32 nn[C::L];
    load_modulus<C>(nn, nmod, ln);
    size_t el = (size_t)blockIdx.x * C::EPB + ln.eslot;
    bool live = el < n;
    size_t ec = live ? el : n - 1;
    u32 a[C::L];
    load_elem<C>(a, in + ec * C::W, ln);
    small_to_lds<C>(ln, 1u);
    u32 r[C::L];
    mont_mul<C>(r, a, ln, nn, n0inv);
    canonicalize<C>(r, nn, ln);
    u32* dst = out + ec * NW;
    emit_words<C, NW>(r, ln, [&](int k, u32 w) { if (live) dst[k] = w; });
}

// ---------------------------------------------------------------------------------------------
// K4: out[i] = x[i] * y[i]        (ystride = 0: every x[i] times the single element y)
// ---------------------------------------------------------------------------------------------
template <class C>
__global__ void __launch_bounds__(BLOCK, C::MINW)
k_mul(u32* __restrict__ out, const u32* __restrict__ x, const u32* __restrict__ y, size_t ystride, size_t n,
      const u32* __restrict__ nmod, u32 n0inv) {
    extern __shared__ u32 lds[];
    Lane<C> ln(lds);
    u32 nn[C::L];
    load_modulus<C>(nn, nmod, ln);
    size_t el = (size_t)blockIdx.x * C::EPB + ln.eslot;
    bool live = el < n;
    size_t ec = live ? el : n - 1;
    u32 a[C::L];
    load_elem<C>(a, x + ec * C::W, ln);
    load_elem_to_lds<C>(ln, y + ec * ystride);
    u32 r[C::L];
    mont_mul<C>(r, a, ln, nn, n0inv);
    canonicalize<C>(r, nn, ln);
    if (live) store_elem<C>(out + el * C::W, r, ln);
}

// ---------------------------------------------------------------------------------------------
// K1a / K1b: out[i] = x[i] ^ e[i]   fixed-window (wbits), left to right, per-element exponents.
//   e: packed little-endian words, element i at e + i*estride (estride = 0: one shared exponent)
//   tab: scratch of gridDim.x*EPB*(2^wbits)*W words: the element's table of x^0..x^(2^w-1), one
//        contiguous row per entry (gathered with 16-byte loads).
// Persistent grid: a workgroup loops over tiles of EPB elements.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 exp_digit(const u32* __restrict__ ep, int ewords, int pos, int wbits) {
    int k = pos >> 5, sh = pos & 31;
    u32 lo = k < ewords ? ep[k] : 0;
    u32 hi = k + 1 < ewords ? ep[k + 1] : 0;
    u64 both = ((u64)hi << 32) | lo;
    return (u32)(both >> sh) & ((1u << wbits) - 1);
}

template <class C>
__global__ void __launch_bounds__(BLOCK, C::MINW)
k_modpow(u32* __restrict__ out, const u32* __restrict__ x, const u32* __restrict__ e, int ewords, size_t estride,
         int ebits, int wbits, size_t n, const u32* __restrict__ nmod, u32 n0inv, const u32* __restrict__ one_m,
         u32* __restrict__ tab) {
    constexpr int W = C::W;
    extern __shared__ u32 lds[];
    Lane<C> ln(lds);
    u32 nn[C::L];
    load_modulus<C>(nn, nmod, ln);
    const size_t ntiles = (n + C::EPB - 1) / C::EPB;
    const int tsize = 1 << wbits;
    u32* mytab = tab + ((size_t)blockIdx.x * C::EPB + ln.eslot) * (size_t)tsize * W;
    const int nwin = (ebits + wbits - 1) / wbits;

    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        size_t el = t * C::EPB + ln.eslot;
        bool live = el < n;
        size_t ec = live ? el : n - 1;
        const u32* ep = e + ec * estride;
        u32 a[C::L];
        // table: tab[0] = 1, tab[1] = x, tab[k] = tab[k-1] * x
        load_elem<C>(a, x + ec * W, ln);
        {
            u32 o[C::L];
            load_modulus<C>(o, one_m, ln);
            store_elem<C>(mytab, o, ln);
        }
        store_elem<C>(mytab + W, a, ln);
        regs_to_lds<C>(ln, a);
#pragma unroll 1
        for (int k = 2; k < tsize; ++k) {
            u32 r[C::L];
            mont_mul<C>(r, a, ln, nn, n0inv);          // x * tab[k-1]
            store_elem<C>(mytab + (size_t)k * W, r, ln);
            regs_to_lds<C>(ln, r);
        }
        // main loop
        u32 d = exp_digit(ep, ewords, (nwin - 1) * wbits, wbits);
        load_elem<C>(a, mytab + (size_t)d * W, ln);
#pragma unroll 1
        for (int wi = nwin - 2; wi >= 0; --wi) {
#pragma unroll 1
            for (int s = 0; s < wbits; ++s) {
                regs_to_lds<C>(ln, a);
                mont_sqr<C>(a, a, ln, nn, n0inv);
            }
            d = exp_digit(ep, ewords, wi * wbits, wbits);
            load_elem_to_lds<C>(ln, mytab + (size_t)d * W);
            mont_mul<C>(a, a, ln, nn, n0inv);
        }
        canonicalize<C>(a, nn, ln);
        if (live) store_elem<C>(out + el * W, a, ln);
    }
}

// ---------------------------------------------------------------------------------------------
// k_modpow for arrays of MORE than one round of tiles, in phases.  Every tile of k_modpow takes the same time T, so an
// array of r = ntiles / slots rounds finishes after ceil(r) T: 10^6 elements are 7.63 rounds of the 512 workgroup slots and
// cost 8 -- 4.6 % of the launch is the idle tail of its last round (measured: 0.632 of the roof at exactly 2 rounds,
// 0.603 at 7.63).  Here a tile's power is cut into `phases` runs of windows and the workgroups take (phase, tile) units
// from a queue in phase-major order: P x 7.63 rounds of units of T / P each, the tail is at most one UNIT.  Between its
// phases a tile's running value lives in out[], its window table in a table of its OWN (per element, not per lane slot:
// another workgroup continues it).  Unit u = (phase, tile) needs (phase - 1, tile), which is unit u - ntiles: handed out
// earlier, to a workgroup that is therefore running and waits for nothing later than itself -- the spin below always ends,
// whatever part of the grid is resident.  Same products in the same order as k_modpow: bit-identical results.
// ---------------------------------------------------------------------------------------------
template <class C>
__global__ void __launch_bounds__(BLOCK, C::MINW)
k_modpow_phased(u32* __restrict__ out, const u32* __restrict__ x, const u32* __restrict__ e, int ewords, size_t estride,
                int ebits, int wbits, size_t n, const u32* __restrict__ nmod, u32 n0inv, const u32* __restrict__ one_m,
                u32* __restrict__ tab, int phases, u32* __restrict__ queue, u32* __restrict__ done) {
    constexpr int W = C::W;
    extern __shared__ u32 lds[];
    __shared__ u32 s_unit;
    Lane<C> ln(lds);
    u32 nn[C::L];
    load_modulus<C>(nn, nmod, ln);
    const u32 ntiles = (u32)((n + C::EPB - 1) / C::EPB);
    const u32 nunits = ntiles * (u32)phases;
    const int tsize = 1 << wbits;
    const int nwin = (ebits + wbits - 1) / wbits;
    const int M = nwin - 1;                              // windows of the main loop (the top one is the first table read)
    // ONE thread-0 region per turn -- the hand-over of the finished unit and the fetch of the next, between two barriers.  (With
    // the fetch at the top of the loop and the hand-over at its bottom, two thread-0 regions sit around the back edge: the
    // compiler lets the other lanes of wave 0 run ahead into the next turn's barrier while lane 0 is still signalling, the
    // barrier counts go out of step and the kernel hangs -- tools/micro/queue_handoff.hip reproduces both forms.)
    if (threadIdx.x == 0) s_unit = atomicAdd(queue, 1u);
    __syncthreads();
    for (;;) {
        const u32 u = (u32)__builtin_amdgcn_readfirstlane((int)s_unit);
        if (u >= nunits) break;
        const int ph = (int)(u / ntiles);
        const u32 t = u - (u32)ph * ntiles;
        if (ph > 0) {                                    // the tile's previous phase, run by another workgroup
            if (threadIdx.x == 0) {
                long spins = 0;                          // (a unit lasts milliseconds: 2^28 polls are minutes -- a bug, and then a trap, not a hang)
                while (__hip_atomic_load(done + t, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (u32)ph) {
                    if (++spins > (1L << 28)) __builtin_trap();
                    __builtin_amdgcn_s_sleep(16);
                }
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        size_t el = (size_t)t * C::EPB + ln.eslot;
        bool live = el < n;
        size_t ec = live ? el : n - 1;
        const u32* ep = e + ec * estride;
        u32* mytab = tab + ((size_t)t * C::EPB + ln.eslot) * (size_t)tsize * W;
        u32 a[C::L];
        if (ph == 0) {
            // table: tab[0] = 1, tab[1] = x, tab[k] = tab[k-1] * x
            load_elem<C>(a, x + ec * W, ln);
            {
                u32 o[C::L];
                load_modulus<C>(o, one_m, ln);
                store_elem<C>(mytab, o, ln);
            }
            store_elem<C>(mytab + W, a, ln);
            regs_to_lds<C>(ln, a);
#pragma unroll 1
            for (int k = 2; k < tsize; ++k) {
                u32 r[C::L];
                mont_mul<C>(r, a, ln, nn, n0inv);          // x * tab[k-1]
                store_elem<C>(mytab + (size_t)k * W, r, ln);
                regs_to_lds<C>(ln, r);
            }
            u32 d = exp_digit(ep, ewords, (nwin - 1) * wbits, wbits);
            load_elem<C>(a, mytab + (size_t)d * W, ln);
        } else {
            load_elem<C>(a, out + ec * W, ln);
        }
        // the windows of this phase: M - 1 - M ph / P  down to  M - M (ph + 1) / P
        const int hi = M - 1 - (int)((long)M * ph / phases), lo = M - (int)((long)M * (ph + 1) / phases);
#pragma unroll 1
        for (int wi = hi; wi >= lo; --wi) {
#pragma unroll 1
            for (int s = 0; s < wbits; ++s) {
                regs_to_lds<C>(ln, a);
                mont_sqr<C>(a, a, ln, nn, n0inv);
            }
            u32 d = exp_digit(ep, ewords, wi * wbits, wbits);
            load_elem_to_lds<C>(ln, mytab + (size_t)d * W);
            mont_mul<C>(a, a, ln, nn, n0inv);
        }
        if (ph == phases - 1) canonicalize<C>(a, nn, ln);
        if (live) store_elem<C>(out + el * W, a, ln);
        const bool hand_on = ph < phases - 1;            // another workgroup continues this tile
        if (hand_on) __threadfence();
        __syncthreads();                                 // every store of the unit is out; everybody has read s_unit
        if (threadIdx.x == 0) {
            if (hand_on) (void)__hip_atomic_exchange(done + t, (u32)(ph + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            s_unit = atomicAdd(queue, 1u);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Simultaneous power of two bases (Straus): out[i] = x[i]^e1 * y[i]^e2 with the squarings shared -- max(ebits1, ebits2)
// squarings instead of their sum.  Either exponent may be shared (stride 0) or per element.  Per-lane tables of both bases
// in scratch (2 * 2^w rows).  Used by the verifiers' check (B) in the form B_i^v (B_{i-1}^{-1})^{k_E,i}.
// ---------------------------------------------------------------------------------------------
template <class C>
__global__ void __launch_bounds__(BLOCK, C::MINW)
k_modpow2(u32* __restrict__ out, const u32* __restrict__ x, const u32* __restrict__ e1, int ewords1, size_t estride1, int ebits1,
          const u32* __restrict__ y, const u32* __restrict__ e2, int ewords2, size_t estride2, int ebits2, int wbits, size_t n,
          const u32* __restrict__ nmod, u32 n0inv, const u32* __restrict__ one_m, u32* __restrict__ tab) {
    constexpr int W = C::W;
    extern __shared__ u32 lds[];
    Lane<C> ln(lds);
    u32 nn[C::L];
    load_modulus<C>(nn, nmod, ln);
    const size_t ntiles = (n + C::EPB - 1) / C::EPB;
    const int tsize = 1 << wbits;
    u32* tab1 = tab + ((size_t)blockIdx.x * C::EPB + ln.eslot) * (size_t)(2 * tsize) * W;
    u32* tab2 = tab1 + (size_t)tsize * W;
    const int nwin1 = (ebits1 + wbits - 1) / wbits, nwin2 = (ebits2 + wbits - 1) / wbits;
    const int nwin = nwin1 > nwin2 ? nwin1 : nwin2;

    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        size_t el = t * C::EPB + ln.eslot;
        bool live = el < n;
        size_t ec = live ? el : n - 1;
        const u32* ep1 = e1 + ec * estride1;
        const u32* ep2 = e2 + ec * estride2;
        u32 a[C::L];
        // tables: tab[0] = 1, tab[1] = base, tab[k] = tab[k-1] * base
#pragma unroll 1
        for (int which = 0; which < 2; ++which) {
            u32* tb = which ? tab2 : tab1;
            {
                u32 o[C::L];
                load_modulus<C>(o, one_m, ln);
                store_elem<C>(tb, o, ln);
            }
            load_elem<C>(a, (which ? y : x) + ec * W, ln);
            store_elem<C>(tb + W, a, ln);
            regs_to_lds<C>(ln, a);
#pragma unroll 1
            for (int k = 2; k < tsize; ++k) {
                u32 r[C::L];
                mont_mul<C>(r, a, ln, nn, n0inv);          // base * tab[k-1]
                store_elem<C>(tb + (size_t)k * W, r, ln);
                regs_to_lds<C>(ln, r);
            }
        }
        load_modulus<C>(a, one_m, ln);
#pragma unroll 1
        for (int wi = nwin - 1; wi >= 0; --wi) {
            if (wi != nwin - 1) {
#pragma unroll 1
                for (int s = 0; s < wbits; ++s) {
                    regs_to_lds<C>(ln, a);
                    mont_sqr<C>(a, a, ln, nn, n0inv);
                }
            }
            if (wi < nwin1) {                              // (uniform over the launch: window counts, not digits, decide)
                u32 d = exp_digit(ep1, ewords1, wi * wbits, wbits);
                load_elem_to_lds<C>(ln, tab1 + (size_t)d * W);
                mont_mul<C>(a, a, ln, nn, n0inv);
            }
            if (wi < nwin2) {
                u32 d = exp_digit(ep2, ewords2, wi * wbits, wbits);
                load_elem_to_lds<C>(ln, tab2 + (size_t)d * W);
                mont_mul<C>(a, a, ln, nn, n0inv);
            }
        }
        canonicalize<C>(a, nn, ln);
        if (live) store_elem<C>(out + el * W, a, ln);
    }
}

// k_modpow2 for arrays of more than one round of tiles: the same simultaneous power in phases from a queue of (phase, tile)
// units (see k_modpow_phased -- the queue, the hand-over and the ONE thread-0 region per turn are the same); both tables
// of a tile live in a table of its own (2 * 2^w rows per element).
template <class C>
__global__ void __launch_bounds__(BLOCK, C::MINW)
k_modpow2_phased(u32* __restrict__ out, const u32* __restrict__ x, const u32* __restrict__ e1, int ewords1, size_t estride1, int ebits1,
                 const u32* __restrict__ y, const u32* __restrict__ e2, int ewords2, size_t estride2, int ebits2, int wbits, size_t n,
                 const u32* __restrict__ nmod, u32 n0inv, const u32* __restrict__ one_m, u32* __restrict__ tab, int phases,
                 u32* __restrict__ queue, u32* __restrict__ done) {
    constexpr int W = C::W;
    extern __shared__ u32 lds[];
    __shared__ u32 s_unit;
    Lane<C> ln(lds);
    u32 nn[C::L];
    load_modulus<C>(nn, nmod, ln);
    const u32 ntiles = (u32)((n + C::EPB - 1) / C::EPB);
    const u32 nunits = ntiles * (u32)phases;
    const int tsize = 1 << wbits;
    const int nwin1 = (ebits1 + wbits - 1) / wbits, nwin2 = (ebits2 + wbits - 1) / wbits;
    const int nwin = nwin1 > nwin2 ? nwin1 : nwin2;
    if (threadIdx.x == 0) s_unit = atomicAdd(queue, 1u);
    __syncthreads();
    for (;;) {
        const u32 u = (u32)__builtin_amdgcn_readfirstlane((int)s_unit);
        if (u >= nunits) break;
        const int ph = (int)(u / ntiles);
        const u32 t = u - (u32)ph * ntiles;
        if (ph > 0) {
            if (threadIdx.x == 0) {
                long spins = 0;
                while (__hip_atomic_load(done + t, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (u32)ph) {
                    if (++spins > (1L << 28)) __builtin_trap();
                    __builtin_amdgcn_s_sleep(16);
                }
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        size_t el = (size_t)t * C::EPB + ln.eslot;
        bool live = el < n;
        size_t ec = live ? el : n - 1;
        const u32* ep1 = e1 + ec * estride1;
        const u32* ep2 = e2 + ec * estride2;
        u32* tab1 = tab + ((size_t)t * C::EPB + ln.eslot) * (size_t)(2 * tsize) * W;
        u32* tab2 = tab1 + (size_t)tsize * W;
        u32 a[C::L];
        if (ph == 0) {
            // tables: tab[0] = 1, tab[1] = base, tab[k] = tab[k-1] * base
#pragma unroll 1
            for (int which = 0; which < 2; ++which) {
                u32* tb = which ? tab2 : tab1;
                {
                    u32 o[C::L];
                    load_modulus<C>(o, one_m, ln);
                    store_elem<C>(tb, o, ln);
                }
                load_elem<C>(a, (which ? y : x) + ec * W, ln);
                store_elem<C>(tb + W, a, ln);
                regs_to_lds<C>(ln, a);
#pragma unroll 1
                for (int k = 2; k < tsize; ++k) {
                    u32 r[C::L];
                    mont_mul<C>(r, a, ln, nn, n0inv);          // base * tab[k-1]
                    store_elem<C>(tb + (size_t)k * W, r, ln);
                    regs_to_lds<C>(ln, r);
                }
            }
            load_modulus<C>(a, one_m, ln);
        } else {
            load_elem<C>(a, out + ec * W, ln);
        }
        // the windows of this phase: nwin - 1 - nwin ph / P  down to  nwin - nwin (ph + 1) / P
        const int hi = nwin - 1 - (int)((long)nwin * ph / phases), lo = nwin - (int)((long)nwin * (ph + 1) / phases);
#pragma unroll 1
        for (int wi = hi; wi >= lo; --wi) {
            if (wi != nwin - 1) {
#pragma unroll 1
                for (int s = 0; s < wbits; ++s) {
                    regs_to_lds<C>(ln, a);
                    mont_sqr<C>(a, a, ln, nn, n0inv);
                }
            }
            if (wi < nwin1) {
                u32 d = exp_digit(ep1, ewords1, wi * wbits, wbits);
                load_elem_to_lds<C>(ln, tab1 + (size_t)d * W);
                mont_mul<C>(a, a, ln, nn, n0inv);
            }
            if (wi < nwin2) {
                u32 d = exp_digit(ep2, ewords2, wi * wbits, wbits);
                load_elem_to_lds<C>(ln, tab2 + (size_t)d * W);
                mont_mul<C>(a, a, ln, nn, n0inv);
            }
        }
        if (ph == phases - 1) canonicalize<C>(a, nn, ln);
        if (live) store_elem<C>(out + el * W, a, ln);
        const bool hand_on = ph < phases - 1;
        if (hand_on) __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            if (hand_on) (void)__hip_atomic_exchange(done + t, (u32)(ph + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            s_unit = atomicAdd(queue, 1u);
        }
        __syncthreads();
    }
}

// =============================================================================================
// second part: fixed-base tables (K2), multi-exponentiation (K3), reductions (K5), comparison
// (K6), data movement (K7) and the ring kernels over Z_q (K8).
// =============================================================================================

// ---------------------------------------------------------------------------------------------
// K5 and friends: strided reduction.  out[seg][j] = OP_k x[seg][j + k*L], k = 0 .. ceil(len/L)-1,
// for j < L.  OP = Montgomery product (MUL = true) or modular sum.  nseg segments of `len`
// elements each; the output has nseg segments of L elements.
// ---------------------------------------------------------------------------------------------
template <class C, bool MUL>
__global__ void __launch_bounds__(BLOCK, C::MINW)
k_reduce_strided(u32* __restrict__ out, const u32* __restrict__ x, size_t len, size_t Lout, size_t nseg,
                 const u32* __restrict__ nmod, u32 n0inv) {
    constexpr int W = C::W;
    extern __shared__ u32 lds[];
    Lane<C> ln(lds);
    u32 nn[C::L];
    load_modulus<C>(nn, nmod, ln);
    size_t t = (size_t)blockIdx.x * C::EPB + ln.eslot;
    bool live = t < nseg * Lout;
    size_t tc = live ? t : nseg * Lout - 1;
    size_t seg = tc / Lout, j = tc % Lout;
    const u32* base = x + seg * len * W;
    u32 acc[C::L];
    load_elem<C>(acc, base + j * W, ln);
    size_t cnt = (len - j + Lout - 1) / Lout;            // elements j, j+Lout, ... < len
    for (size_t k = 1; k < cnt; ++k) {
        const u32* src = base + (j + k * Lout) * W;
        if constexpr (MUL) {
            load_elem_to_lds<C>(ln, src);
            mont_mul<C>(acc, acc, ln, nn, n0inv);
        } else {
            u32 b[C::L];
            load_elem<C>(b, src, ln);
            mod_add<C>(acc, acc, b, nn, ln);
        }
    }
    if constexpr (MUL) canonicalize<C>(acc, nn, ln);
    if (live) store_elem<C>(out + t * W, acc, ln);
}

// ---------------------------------------------------------------------------------------------
// K8 element-wise ring kernels:  op 0: x + y   op 1: -x   op 2: x*v + y (v one element)   op 3: x*v
// ---------------------------------------------------------------------------------------------
template <class C>
__global__ void __launch_bounds__(BLOCK, C::MINW)
k_ring_elementwise(u32* __restrict__ out, const u32* __restrict__ x, const u32* __restrict__ y, const u32* __restrict__ v,
                   int op, size_t n, const u32* __restrict__ nmod, u32 n0inv) {
    constexpr int W = C::W;
    extern __shared__ u32 lds[];
    Lane<C> ln(lds);
    u32 nn[C::L];
    load_modulus<C>(nn, nmod, ln);
    size_t el = (size_t)blockIdx.x * C::EPB + ln.eslot;
    bool live = el < n;
    size_t ec = live ? el : n - 1;
    u32 a[C::L], r[C::L];
    load_elem<C>(a, x + ec * W, ln);
    if (op == 0) {
        u32 b[C::L];
        load_elem<C>(b, y + ec * W, ln);
        mod_add<C>(r, a, b, nn, ln);
    } else if (op == 1) {
        mod_neg<C>(r, a, nn, ln);
    } else {
        load_elem_to_lds<C>(ln, v);
        u32 t[C::L];
        mont_mul<C>(t, a, ln, nn, n0inv);
        canonicalize<C>(t, nn, ln);
        if (op == 2) {
            u32 b[C::L];
            load_elem<C>(b, y + ec * W, ln);
            mod_add<C>(r, t, b, nn, ln);
        } else {
#pragma unroll
            for (int j = 0; j < C::L; ++j) r[j] = t[j];
        }
    }
    if (live) store_elem<C>(out + el * W, r, ln);
}

// ---------------------------------------------------------------------------------------------
// K8 scans.  An affine recurrence x[i] = x[i-1]*e[i] + b[i] (recLin; b == nullptr: prods,
// y[i] = y[i-1]*e[i]) over segments of `seglen` elements (seglen % Cc == 0 or one segment),
// processed in chunks of Cc consecutive elements per lane (pair):
//   k_scan_totals : per chunk, the composed map (E = prod e, X = value reached from 0)
//   (recursion on the totals gives every chunk's incoming value)
//   k_scan_apply  : per chunk, replay the recurrence from the incoming value and store x[i]
// `rev`: element i of a segment is read/written at position seglen-1-i (suffix scans of K3).
// ---------------------------------------------------------------------------------------------
// WANT_X = false: tot[c] = prod of the chunk's e.   WANT_X = true: tot[c] = value reached from 0.
// (Two launches for recLin: one accumulator per kernel keeps a + columns + accumulator within 256 VGPRs.)
template <class C, bool WANT_X>
__global__ void __launch_bounds__(BLOCK, C::MINW)
k_scan_totals(u32* __restrict__ tot, const u32* __restrict__ e, const u32* __restrict__ b,
              size_t n, size_t Cc, size_t seglen, int rev, const u32* __restrict__ nmod, u32 n0inv,
              const u32* __restrict__ one_m) {
    constexpr int W = C::W;
    extern __shared__ u32 lds[];
    Lane<C> ln(lds);
    u32 nn[C::L];
    load_modulus<C>(nn, nmod, ln);
    size_t nchunks = (n + Cc - 1) / Cc;
    size_t c = (size_t)blockIdx.x * C::EPB + ln.eslot;
    bool live = c < nchunks;
    size_t cc = live ? c : nchunks - 1;
    size_t lo = cc * Cc, hi = lo + Cc < n ? lo + Cc : n;
    u32 A[C::L];
    if constexpr (WANT_X) {
#pragma unroll
        for (int j = 0; j < C::L; ++j) A[j] = 0u;
    } else {
        load_modulus<C>(A, one_m, ln);
    }
    for (size_t i = lo; i < lo + Cc; ++i) {          // uniform trip count; short chunks idle at the end
        if (i < hi) {
            size_t pos = rev ? (i / seglen) * seglen + (seglen - 1 - i % seglen) : i;
            load_elem_to_lds<C>(ln, e + pos * W);
            mont_mul<C>(A, A, ln, nn, n0inv);
            if constexpr (WANT_X) {
                canonicalize<C>(A, nn, ln);
                u32 bb[C::L];
                load_elem<C>(bb, b + pos * W, ln);
                mod_add<C>(A, A, bb, nn, ln);
            }
        }
    }
    canonicalize<C>(A, nn, ln);
    if (live) store_elem<C>(tot + c * W, A, ln);
}

// incoming: per-chunk inclusive results of the level above (chunk c starts from incoming[c-1]),
// nullptr = every chunk starts fresh.  A chunk that begins a segment starts fresh (0 / one).
template <class C>
__global__ void __launch_bounds__(BLOCK, C::MINW)
k_scan_apply(u32* __restrict__ out, const u32* __restrict__ e, const u32* __restrict__ b, const u32* __restrict__ incoming,
             size_t n, size_t Cc, size_t seglen, int rev, const u32* __restrict__ nmod, u32 n0inv,
             const u32* __restrict__ one_m) {
    constexpr int W = C::W;
    extern __shared__ u32 lds[];
    Lane<C> ln(lds);
    u32 nn[C::L];
    load_modulus<C>(nn, nmod, ln);
    size_t nchunks = (n + Cc - 1) / Cc;
    size_t c = (size_t)blockIdx.x * C::EPB + ln.eslot;
    bool live = c < nchunks;
    size_t cc = live ? c : nchunks - 1;
    size_t lo = cc * Cc, hi = lo + Cc < n ? lo + Cc : n;
    bool fresh = incoming == nullptr || (lo % seglen) == 0;
    u32 X[C::L];
    if (fresh) {
        if (b) {
#pragma unroll
            for (int j = 0; j < C::L; ++j) X[j] = 0u;
        } else {
            load_modulus<C>(X, one_m, ln);
        }
    } else {
        load_elem<C>(X, incoming + (cc - 1) * W, ln);
    }
    for (size_t i = lo; i < lo + Cc; ++i) {
        if (i < hi) {
            size_t pos = rev ? (i / seglen) * seglen + (seglen - 1 - i % seglen) : i;
            load_elem_to_lds<C>(ln, e + pos * W);
            u32 t[C::L];
            mont_mul<C>(t, X, ln, nn, n0inv);
            canonicalize<C>(t, nn, ln);
            if (b) {
                u32 bb[C::L];
                load_elem<C>(bb, b + pos * W, ln);
                mod_add<C>(X, t, bb, nn, ln);
            } else {
#pragma unroll
                for (int j = 0; j < C::L; ++j) X[j] = t[j];
            }
            if (live) store_elem<C>(out + pos * W, X, ln);
        }
    }
}

template <class C>
__global__ void __launch_bounds__(BLOCK, C::MINW)
k_fixed_level(u32* __restrict__ T, int w, int nwin, int l, const u32* __restrict__ nmod, u32 n0inv) {
    constexpr int W = C::W;
    extern __shared__ u32 lds[];
    Lane<C> ln(lds);
    u32 nn[C::L];
    load_modulus<C>(nn, nmod, ln);
    size_t per = ((size_t)1 << l) - 1;                 // d = 2^l + 1 .. 2^(l+1) - 1
    size_t t = (size_t)blockIdx.x * C::EPB + ln.eslot;
    bool live = t < per * nwin;
    size_t tc = live ? t : per * nwin - 1;
    size_t k = tc / per, r = tc % per + 1;             // r = d - 2^l in [1, 2^l)
    u32* row = T + (k << w) * W;
    u32 a[C::L];
    load_elem<C>(a, row + r * W, ln);
    load_elem_to_lds<C>(ln, row + ((size_t)1 << l) * W);
    u32 o[C::L];
    mont_mul<C>(o, a, ln, nn, n0inv);
    canonicalize<C>(o, nn, ln);
    if (live) store_elem<C>(row + (((size_t)1 << l) + r) * W, o, ln);
}

// out[i] = prod_k T[k][digit_k(e[i])].  parts > 1 (small arrays): the chain of nwin - 1 dependent products of an element is cut
// into `parts` independent pieces -- item t = part * n + i multiplies the windows [part nwin / parts, (part + 1) nwin / parts)
// of element i into out[t] -- and the caller multiplies the pieces together by a tree of element-wise products: a chain of
// nwin / parts + log2(parts) products instead of nwin, on `parts` times as many lanes (which a small array leaves idle anyway).
template <class C>
__global__ void __launch_bounds__(BLOCK, C::MINW)
k_fixed_exp(u32* __restrict__ out, const u32* __restrict__ T, int w, int nwin, const u32* __restrict__ e, int ewords,
            size_t n, int parts, const u32* __restrict__ nmod, u32 n0inv) {
    constexpr int W = C::W;
    extern __shared__ u32 lds[];
    Lane<C> ln(lds);
    u32 nn[C::L];
    load_modulus<C>(nn, nmod, ln);
    const size_t items = n * (size_t)parts;
    const size_t ntiles = (items + C::EPB - 1) / C::EPB;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        size_t it = t * C::EPB + ln.eslot;
        bool live = it < items;
        size_t ic = live ? it : items - 1;
        const int part = (int)(ic / n);
        const size_t ec = ic % n;
        const int k0 = (int)((long)part * nwin / parts), k1 = (int)((long)(part + 1) * nwin / parts);
        const u32* ep = e + ec * ewords;
        u32 a[C::L];
        u32 d = exp_digit(ep, ewords, k0 * w, w);
        load_elem<C>(a, T + (((size_t)k0 << w) + d) * W, ln);
#pragma unroll 1
        for (int k = k0 + 1; k < k1; ++k) {
            d = exp_digit(ep, ewords, k * w, w);
            load_elem_to_lds<C>(ln, T + (((size_t)k << w) + d) * W);
            mont_mul<C>(a, a, ln, nn, n0inv);
        }
        canonicalize<C>(a, nn, ln);
        if (live) store_elem<C>(out + it * W, a, ln);
    }
}

// One level of the per-bucket product tree.  Bucket b owns cnt_in[b] items at off_in[b]; output item
// (b, j) = product of its input items [jF, (j+1)F) and lands at off_out[b] + j.  One lane (pair) per output
// item, so a bucket of any size is spread over ceil(size/F) lanes: no lane ever walks a long bucket
// (skewed digits -- a short top window, equal exponents -- would otherwise serialise on one lane).
// FIRST: input items are rows of x selected through `sorted`; otherwise rows of `in`.
// The arrays of one launch (a multi-exponentiation of k arrays under ONE exponent vector builds k trees of the same shape:
// counts and offsets are shared, so one launch per level serves up to LEVEL_ARRAYS of them -- a seventh of the launches over
// curves, and k times as many items in the thin upper levels).  Block b works for array b / blocks_per_array.
constexpr int LEVEL_ARRAYS = 8;
struct LevelInputs {
    const u32* p[LEVEL_ARRAYS];
};
template <class C, bool FIRST>
__global__ void __launch_bounds__(BLOCK, C::MINW)
k_bucket_level(u32* __restrict__ out, size_t out_stride, LevelInputs ins, unsigned blocks_per_array, const u32* __restrict__ sorted,
               const u32* __restrict__ off_in, const u32* __restrict__ cnt_in, const u32* __restrict__ off_out,
               size_t nbuckets, size_t total_out, u32 F, const u32* __restrict__ nmod, u32 n0inv) {
    constexpr int W = C::W;
    extern __shared__ u32 lds[];
    Lane<C> ln(lds);
    u32 nn[C::L];
    load_modulus<C>(nn, nmod, ln);
    const unsigned arr = blockIdx.x / blocks_per_array;          // (wave-uniform)
    const u32* __restrict__ in = ins.p[arr];
    out += (size_t)arr * out_stride;
    size_t t = (size_t)(blockIdx.x % blocks_per_array) * C::EPB + ln.eslot;
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
    u32 acc[C::L];
    auto row = [&](u32 k) -> const u32* { return FIRST ? in + (size_t)sorted[k] * W : in + (size_t)k * W; };
    load_elem<C>(acc, row(start), ln);
    for (u32 k = start + 1; k < end; ++k) {
        load_elem_to_lds<C>(ln, row(k));
        mont_mul<C>(acc, acc, ln, nn, n0inv);
    }
    canonicalize<C>(acc, nn, ln);
    if (live) store_elem<C>(out + t * W, acc, ln);
}

// ---------------------------------------------------------------------------------------------
// K10: membership in the order-q subgroup of a safe-prime group (p = 2q + 1): x is a member iff it is a quadratic
// residue iff the Jacobi symbol (x / p) = 1.  Binary Jacobi algorithm on 28-bit limbs, one element per lane (moduli
// up to 3072 bits), no multiplications: a few limb passes per step, at most 2 * bits(p) steps -- about a tenth of
// the instructions of the x^q = 1 test.  The rows are in Montgomery form x R mod p; (R / p) = 1 because R is an even
// power of two, so the symbol of the row is the symbol of x.  flags[0] |= 1 when some element is not a member
// (zero included).  ref: the membership test VCR makes when an array is read (pGroup.toElementArray), e.g.
// P/hvzk/PoSBasicTW.java:787-792.
// ---------------------------------------------------------------------------------------------
template <class C>
__global__ void __launch_bounds__(BLOCK, 2)
k_jacobi_member(const u32* __restrict__ x, size_t n, const u32* __restrict__ nmod, u32* __restrict__ flags) {
    // ONE lane holds the whole element: a and m in registers (2 S <= 148 VGPRs), updated in place.  (The loads below also
    // read the two-share layout of Cfg<110, 2>; 3072-bit moduli use k_jacobi_member_lanes since that proved three times faster.)
    static_assert(C::LPE <= 2, "the element must fit one lane");
    constexpr int S = C::S, L = C::L, LW = C::LW;
    size_t el = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    bool live = el < n;
    size_t ec = live ? el : n - 1;
    u32 a[S], m[S];
#pragma unroll
    for (int h = 0; h < C::LPE; ++h) {
        const uint4* q = reinterpret_cast<const uint4*>(x + ec * C::W + h * LW);
        const uint4* qm = reinterpret_cast<const uint4*>(nmod + h * LW);
#pragma unroll
        for (int k = 0; k < LW / 4; ++k) {
            uint4 v = q[k], w = qm[k];
            if (4 * k + 0 < L) { a[h * L + 4 * k + 0] = v.x; m[h * L + 4 * k + 0] = w.x; }
            if (4 * k + 1 < L) { a[h * L + 4 * k + 1] = v.y; m[h * L + 4 * k + 1] = w.y; }
            if (4 * k + 2 < L) { a[h * L + 4 * k + 2] = v.z; m[h * L + 4 * k + 2] = w.z; }
            if (4 * k + 3 < L) { a[h * L + 4 * k + 3] = v.w; m[h * L + 4 * k + 3] = w.w; }
        }
    }
    u32 t = 0;                                   // parity of the sign flips: symbol = (-1)^t
    u32 nz = 0;
#pragma unroll
    for (int j = 0; j < S; ++j) nz |= a[j];
    bool zero_in = nz == 0;
    // invariant: m odd, 0 <= a < m at loop entry (a < p on input)
    while (nz != 0) {
        // strip the factors of two of a: (2 / m) = -1 iff m = 3, 5 mod 8
        u32 low = a[0];
        if (low == 0) {                          // a whole zero limb: shift by one limb (28 bits: even count, no flip)
#pragma unroll
            for (int j = 0; j + 1 < S; ++j) a[j] = a[j + 1];
            a[S - 1] = 0;
            continue;
        }
        int k = __builtin_ctz(low);
        if (k) {
            u32 m8 = m[0] & 7u;
            if ((k & 1) && (m8 == 3u || m8 == 5u)) t ^= 1u;
#pragma unroll
            for (int j = 0; j < S; ++j) {
                u32 hi = j + 1 < S ? a[j + 1] : 0u;
                a[j] = ((a[j] >> k) | (hi << (LIMB_BITS - k))) & LIMB_MASK;
            }
        }
        // a odd now: compare (borrow of a - m), then subtract in place in the right direction
        int32_t borrow = 0;
#pragma unroll
        for (int j = 0; j < S; ++j) borrow = ((int32_t)a[j] - (int32_t)m[j] + borrow) >> LIMB_BITS;
        if (borrow == 0) {                       // a >= m: a := a - m (even; the symbol is unchanged)
            int32_t c = 0;
#pragma unroll
            for (int j = 0; j < S; ++j) {
                int32_t v = (int32_t)a[j] - (int32_t)m[j] + c;
                a[j] = (u32)v & LIMB_MASK;
                c = v >> LIMB_BITS;
            }
        } else {                                 // a < m: swap by reciprocity, (a, m) := (m - a, a)
            if ((a[0] & 3u) == 3u && (m[0] & 3u) == 3u) t ^= 1u;
            int32_t c = 0;
#pragma unroll
            for (int j = 0; j < S; ++j) {
                u32 old = a[j];
                int32_t v = (int32_t)m[j] - (int32_t)old + c;
                a[j] = (u32)v & LIMB_MASK;
                c = v >> LIMB_BITS;
                m[j] = old;
            }
        }
        nz = 0;
#pragma unroll
        for (int j = 0; j < S; ++j) nz |= a[j];
    }
    // a = 0: the symbol is (-1)^t when m = 1 (coprime), 0 otherwise
    u32 rest = m[0] ^ 1u;
#pragma unroll
    for (int j = 1; j < S; ++j) rest |= m[j];
    bool member = !zero_in && rest == 0 && t == 0;
    if (live && !member) atomicOr(flags, 1u);
}

// The same test with the element spread over LPE lanes (4096-bit moduli: a and m do not fit one lane's registers).  The
// lanes of an element run the binary algorithm in lockstep: every decision is taken from lane 0's low limbs (broadcast),
// the right shifts pull the next lane's low limb in (DPP), comparison and subtraction are the element-wide borrow chains
// of canonicalize() (sub_full: LPE sweeps).  ~1200 instructions per step, at most 2 * bits(p) steps: a tenth of x^q.
template <class C>
__global__ void __launch_bounds__(BLOCK, 2)
k_jacobi_member_lanes(const u32* __restrict__ x, size_t n, const u32* __restrict__ nmod, u32* __restrict__ flags) {
    static_assert(C::LPE > 1 && !C::WIDE, "one lane per element uses k_jacobi_member");
    constexpr int L = C::L, LPE = C::LPE;
    Lane<C> ln(nullptr);
    size_t el = (size_t)blockIdx.x * C::EPB + ln.eslot;
    bool live = el < n;
    size_t ec = live ? el : n - 1;
    u32 a[L], m[L], d[L];
    load_elem<C>(a, x + ec * C::W, ln);
    load_modulus<C>(m, nmod, ln);
    u32 t = 0;
    auto any = [&](const u32 (&v)[L]) {
        u32 nz = 0;
#pragma unroll
        for (int j = 0; j < L; ++j) nz |= v[j];
        return or_all<LPE>(nz);
    };
    u32 nz = any(a);
    const bool zero_in = nz == 0;
    // invariant: m odd, 0 <= a < m at loop entry (a < p on input)
    while (nz != 0) {
        const u32 low = from_lane0<LPE>(a[0]);
        const u32 up = from_above<LPE>(a[0]) & ln.nottopmask;       // the limb above this lane's share (0 above the element)
        if (low == 0) {                          // a whole zero limb: shift by one limb (28 bits: even count, no flip)
#pragma unroll
            for (int j = 0; j + 1 < L; ++j) a[j] = a[j + 1];
            a[L - 1] = up;
            continue;
        }
        const int k = __builtin_ctz(low);
        if (k) {
            const u32 m8 = from_lane0<LPE>(m[0]) & 7u;
            if ((k & 1) && (m8 == 3u || m8 == 5u)) t ^= 1u;
#pragma unroll
            for (int j = 0; j < L; ++j) {
                const u32 hi = j + 1 < L ? a[j + 1] : up;
                a[j] = ((a[j] >> k) | (hi << (LIMB_BITS - k))) & LIMB_MASK;
            }
        }
        // a odd now: a - m over the whole element; borrow = -1 when a < m
        if (sub_full<C>(d, a, m, ln) == 0) {     // a >= m: a := a - m (even; the symbol is unchanged)
#pragma unroll
            for (int j = 0; j < L; ++j) a[j] = d[j];
        } else {                                 // a < m: swap by reciprocity, (a, m) := (m - a, a)
            const u32 a0 = from_lane0<LPE>(a[0]), m0 = from_lane0<LPE>(m[0]);
            if ((a0 & 3u) == 3u && (m0 & 3u) == 3u) t ^= 1u;
            sub_full<C>(d, m, a, ln);
#pragma unroll
            for (int j = 0; j < L; ++j) {
                m[j] = a[j];
                a[j] = d[j];
            }
        }
        nz = any(a);
    }
    // a = 0: the symbol is (-1)^t when m = 1 (coprime), 0 otherwise
    u32 rest = m[0] ^ (ln.half == 0 ? 1u : 0u);
#pragma unroll
    for (int j = 1; j < L; ++j) rest |= m[j];
    rest = or_all<LPE>(rest);
    const bool member = !zero_in && rest == 0 && t == 0;
    if (live && ln.half == 0 && !member) atomicOr(flags, 1u);
}

}  // namespace vmn
