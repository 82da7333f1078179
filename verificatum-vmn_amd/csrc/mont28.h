// mont28.h — device-side big-number core for gfx950 (MI355X).
//
// Representation ("M28 form"): an element x of Z_N is stored as x*R mod N (Montgomery form,
// R = 2^(28*S)), canonical (< N), in S limbs of 28 bits, one limb per 32-bit word.
//
// Why radix 2^28 and not 2^32: on gfx950 v_mad_u64_u32 issues at the same rate as a plain integer
// add (measured, tools/valu_rate.hip -> profiles/valu_rate_r01.txt: ~4.3 cycles per wave64
// instruction per SIMD for both), so the cost of a multiplication is its *instruction count*.
// With 28-bit limbs every 56-bit partial product is accumulated into a 64-bit column with a
// single v_mad_u64_u32 and no carry fix-up: a column receives at most 2*S products (< 2^56 each,
// 2*S < 256), so it cannot overflow.  Carries are resolved once per multiplication.
//
// One element per lane: `a` lives in VGPRs (static indices), the columns T live in VGPR pairs,
// the multiplier b is streamed limb by limb from LDS (dynamic index), the modulus limbs are
// wave-uniform (SGPRs through scalar loads).  No MFMA: this is integer big-number arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vmn {

typedef uint32_t u32;
typedef uint64_t u64;

constexpr u32 LIMB_BITS = 28;
constexpr u32 LIMB_MASK = (1u << LIMB_BITS) - 1;

// Number of 28-bit limbs needed for an n-bit modulus plus the slack Montgomery needs so that
// inputs < 2N give outputs < 2N without a conditional subtraction (R > 4N).
__host__ __device__ constexpr int limbs_for_bits(int nbits) { return (nbits + 2 + 27) / 28; }
// Device stride (in 32-bit words) of one element: limbs rounded up to a multiple of 4 (16 B).
__host__ __device__ constexpr int stride_for_limbs(int s) { return (s + 3) & ~3; }

// Wave-uniform description of a modulus, read through scalar loads.
struct ModulusDev {
    const u32* n;      // S limbs of N (radix 2^28)
    u32 n0inv;         // -N^{-1} mod 2^28
};

// One "row" of the coarsely integrated operand scanning (CIOS) Montgomery multiplication on the
// lazy columns P[0..S-1]:
//     pass 1:  P[j]   = a[j] * bi + P[j]            (top column starts fresh)
//     m        = P[0] * n0inv mod 2^28
//     pass 2:  P[j-1] = m * N[j] + P[j]             (destination shifted by one column)
// After pass 2 column 0 would be = 0 mod 2^28; its upper part is the carry into the new column 0.
// v_mad_u64_u32 has a destination separate from its addend, so the one-column shift of the
// running sum costs no instruction at all: a row is exactly 2*S multiply-adds + 5 VALU.
// The rows are emitted as asm statements by tools/gen_mont_asm.py (gen/mont_rows.inc): hipcc's
// own allocation of the plain C++ form doubles the live ranges and spills to scratch.
template <int S> __device__ __forceinline__ void mont_row_asm_first(u64 (&P)[S], const u32 (&a)[S], u32 b, const u32 (&n)[S], u32 n0inv);
template <int S> __device__ __forceinline__ void mont_row_asm_next(u64 (&P)[S], const u32 (&a)[S], u32 b, const u32 (&n)[S], u32 n0inv);
// Squaring rows (see the generator): rows are grouped in blocks of SQR_BLK; a row multiplies only the
// columns from its block's first column on, later blocks with the doubled limb.  Round 4: blocks of ONE row (the fully
// triangular schedule) -- rounds 1-3 used blocks of 8, whose 64 in-block products per 8 rows where the triangle needs 36
// were 3 % of a squaring: 706.3 -> 696.1 ms on the headline, measured back to back (profiles/r04_sqr_block_sweep.txt).
constexpr int SQR_BLK = 1;
template <int S> __device__ __forceinline__ void mont_sqr_row_asm_first(u64 (&P)[S], const u32 (&a)[S], u32 b, u32 b2, const u32 (&n)[S], u32 n0inv);
template <int S, int J0> __device__ __forceinline__ void mont_sqr_row_asm(u64 (&P)[S], const u32 (&a)[S], u32 b, u32 b2, const u32 (&n)[S], u32 n0inv);
// LPE = 2 or 4 lanes per element (see the generator): L limbs per lane; lowmask = all ones on lane 0 of the
// element, nottopmask = all ones on every lane of the element but the last.
template <int L, int LPE> __device__ __forceinline__ void mont_lanes_row_asm_first(u64 (&P)[L], const u32 (&a)[L], u32 b, const u32 (&n)[L], u32 n0inv, u32 lowmask, u32 nottopmask);
template <int L, int LPE> __device__ __forceinline__ void mont_lanes_row_asm_next(u64 (&P)[L], const u32 (&a)[L], u32 b, const u32 (&n)[L], u32 n0inv, u32 lowmask, u32 nottopmask);
// Squaring rows of the multi-lane geometries (see the generator: every lane skips the same LOCAL columns).
template <int L, int LPE> __device__ __forceinline__ void mont_lanes_sqr_row_asm_first(u64 (&P)[L], const u32 (&a)[L], u32 b, u32 b2, const u32 (&n)[L], u32 n0inv, u32 lowmask, u32 nottopmask);
template <int L, int LPE, int J0> __device__ __forceinline__ void mont_lanes_sqr_row_asm(u64 (&P)[L], const u32 (&a)[L], u32 b, u32 b2, const u32 (&n)[L], u32 n0inv, u32 lowmask, u32 nottopmask);
#include "gen/mont_rows.inc"

// T (S lazy columns, value < 2N when a, b < 2N and R > 4N) = a * b / R mod N.
// b is read as b_lds[i * bstride] (one 28-bit limb per row), n[] are the wave-uniform modulus
// limbs (SGPRs).  Column S-1 is not produced (it is zero: the value lives in columns 0..S-2 plus
// their carries).
template <int S>
__device__ __forceinline__ void mont_mul_columns(u64 (&T)[S], const u32 (&a)[S], const u32* b_lds, int bstride,
                                                 const u32 (&n)[S], u32 n0inv) {
    u32 bi = b_lds[0];
    u32 bn = b_lds[bstride];
    mont_row_asm_first<S>(T, a, bi, n, n0inv);
#pragma unroll 2
    for (int i = 2; i <= S; ++i) {
        bi = bn;
        bn = b_lds[(i < S ? i : 0) * bstride];          // prefetch the next limb under this row
        mont_row_asm_next<S>(T, a, bi, n, n0inv);
    }
    T[S - 1] = 0;
}

// T = a * a / R mod N with the symmetric cross products formed once: the rows of block J0 skip the
// columns before J0 (25 % fewer multiply-adds than the general product at S = 74).  a_lds holds a copy
// of a (the row's own limb needs a dynamic index).
template <int S, int J0>
__device__ __forceinline__ void mont_sqr_blocks(u64 (&T)[S], const u32 (&a)[S], const u32* a_lds, int bstride,
                                                const u32 (&n)[S], u32 n0inv, u32& bn) {
    constexpr int END = J0 + SQR_BLK < S ? J0 + SQR_BLK : S;
#pragma unroll 8          // the whole block (measured back to back on one box: unroll 1 / 2 / 4 / 8 = 751 / 728 / 723 / 718 ms)
    for (int i = (J0 == 0 ? 1 : J0); i < END; ++i) {
        u32 bi = bn;
        bn = a_lds[(i + 1 < S ? i + 1 : 0) * bstride];            // prefetch the next row's limb under this row
        mont_sqr_row_asm<S, J0>(T, a, bi, bi << 1, n, n0inv);
    }
    if constexpr (END < S) mont_sqr_blocks<S, END>(T, a, a_lds, bstride, n, n0inv, bn);
}
template <int S>
__device__ __forceinline__ void mont_sqr_columns(u64 (&T)[S], const u32 (&a)[S], const u32* a_lds, int bstride,
                                                 const u32 (&n)[S], u32 n0inv) {
    u32 b0 = a_lds[0];
    u32 bn = a_lds[bstride];
    mont_sqr_row_asm_first<S>(T, a, b0, b0 << 1, n, n0inv);
    mont_sqr_blocks<S, 0>(T, a, a_lds, bstride, n, n0inv, bn);
    T[S - 1] = 0;
}

// Value of the same register on the lane below within the element (lane h-1; lane 0 receives its own value: mask it).
// (eight lanes per element = half a DPP row of 16 lanes: row_shr:1; lane 0 of an element then reads its neighbour's last
// lane or zero -- masked by the caller like the quad forms.)
template <int LPE>
__device__ __forceinline__ u32 lane_below(u32 x) {
    static_assert(LPE == 2 || LPE == 4 || LPE == 8 || LPE == 16, "lanes per element");
    if constexpr (LPE == 2) return (u32)__builtin_amdgcn_mov_dpp((int)x, 0xA0, 0xf, 0xf, true);     // quad_perm [0,0,2,2]
    else if constexpr (LPE == 4) return (u32)__builtin_amdgcn_mov_dpp((int)x, 0x90, 0xf, 0xf, true);   // quad_perm [0,0,1,2]
    else return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);                 // row_shr:1 (8 or 16 lanes)
}

// Multi-lane product: each lane holds L of the S = LPE*L columns; b_lds streams all S limbs of the multiplier
// (all lanes of an element read the same word).  n[] = this lane's share of the modulus (VGPRs).
// A column collects 2 products < 2^56 per row: 2 S <= 256 rows fit 64 bits (S = 110); for S = 148 the columns are
// relieved once, half way (carry of each column into the next, across lanes through DPP) -- the value is
// unchanged, only its spread over the columns.
template <int L, int LPE>
__device__ __forceinline__ void relieve_columns(u64 (&T)[L], u32 lowmask, u32 nottopmask) {
#pragma unroll
    for (int j = 0; j + 1 < L; ++j) {
        T[j + 1] += T[j] >> LIMB_BITS;
        T[j] &= (u64)LIMB_MASK;
    }
    const u64 top = T[L - 1];
    const u64 cout = top >> LIMB_BITS;
    const u64 nt = ((u64)nottopmask << 32) | nottopmask;
    T[L - 1] = (top & (u64)LIMB_MASK & nt) | (top & ~nt);           // the element's last lane keeps its top column whole
    const u32 lo = lane_below<LPE>((u32)cout) & ~lowmask, hi = lane_below<LPE>((u32)(cout >> 32)) & ~lowmask;
    T[0] += ((u64)hi << 32) | lo;
}
// ROWS <= S = LPE * L is the number of reduction rows (R = 2^(28 ROWS)): the limbs of the multiplier above ROWS are zero
// (the wide geometries of modp_kernels.h), their rows are not run.
// A column collects two products < 2^56 per row: beyond 128 rows (S = 148 and the 8- / 16-lane geometries of the moduli
// above 4096 bits) the columns are relieved every RELIEF_STEP = 74 rows.
constexpr int RELIEF_STEP = 74;
template <int L, int LPE, int ROWS = LPE * L>
__device__ __forceinline__ void mont_mul_columns_lanes(u64 (&T)[L], const u32 (&a)[L], const u32* b_lds, int bstride,
                                                       const u32 (&n)[L], u32 n0inv, u32 lowmask, u32 nottopmask) {
    constexpr int S = ROWS;
    constexpr bool RELIEVE = 2 * S > 256;
    static_assert(ROWS <= LPE * L, "rows beyond the columns");
    u32 bi = b_lds[0];
    u32 bn = b_lds[bstride];
    mont_lanes_row_asm_first<L, LPE>(T, a, bi, n, n0inv, lowmask, nottopmask);
    if constexpr (!RELIEVE) {
#pragma unroll 2
        for (int i = 2; i <= S; ++i) {
            bi = bn;
            bn = b_lds[(i < S ? i : 0) * bstride];
            mont_lanes_row_asm_next<L, LPE>(T, a, bi, n, n0inv, lowmask, nottopmask);
        }
    } else {
#pragma unroll 1
        for (int base = 1; base < S; base += RELIEF_STEP) {             // rows base+1 .. min(base + STEP, S), then a relief
            const int end = base + RELIEF_STEP < S ? base + RELIEF_STEP : S;
#pragma unroll 2
            for (int i = base + 1; i <= end; ++i) {
                bi = bn;
                bn = b_lds[(i < S ? i : 0) * bstride];
                mont_lanes_row_asm_next<L, LPE>(T, a, bi, n, n0inv, lowmask, nottopmask);
            }
            if (end < S) relieve_columns<L, LPE>(T, lowmask, nottopmask);
        }
    }
}

// Multi-lane squaring: T = a^2 / R mod N with every cross product formed once.  The rows whose multiplier limb belongs
// to share u (rows uL .. uL + L - 1) run the block pattern of their LOCAL index on every lane (gen_mont_asm.py,
// gen_pair_sqr): per lane LPE * (L^2 / 2 + L SQR_BLK / 2) products instead of LPE * L^2.  a_lds holds a copy of a.
// LIM = the share's number of rows (L, or what is left of ROWS for the last share of a wide geometry: the limbs above
// are zero, so are all the products the skipped rows would have formed).
template <int L, int LPE, int ROWS, int LIM, int J0>
__device__ __forceinline__ void mont_sqr_lanes_share(u64 (&T)[L], const u32 (&a)[L], const u32* a_lds, int bstride, const u32 (&n)[L],
                                                     u32 n0inv, u32 lowmask, u32 nottopmask, u32& bn, int base, int skip_first) {
    constexpr int END = J0 + SQR_BLK < LIM ? J0 + SQR_BLK : LIM;
#pragma unroll 8
    for (int ip = (J0 == 0 ? skip_first : J0); ip < END; ++ip) {
        const int i = base + ip;
        u32 bi = bn;
        bn = a_lds[(i + 1 < ROWS ? i + 1 : 0) * bstride];         // prefetch the next row's limb under this row
        mont_lanes_sqr_row_asm<L, LPE, J0>(T, a, bi, bi << 1, n, n0inv, lowmask, nottopmask);
    }
    if constexpr (END < LIM) mont_sqr_lanes_share<L, LPE, ROWS, LIM, END>(T, a, a_lds, bstride, n, n0inv, lowmask, nottopmask, bn, base, skip_first);
}
template <int L, int LPE, int ROWS = LPE * L>
__device__ __forceinline__ void mont_sqr_columns_lanes(u64 (&T)[L], const u32 (&a)[L], const u32* a_lds, int bstride,
                                                       const u32 (&n)[L], u32 n0inv, u32 lowmask, u32 nottopmask) {
    constexpr int LAST = ROWS - (LPE - 1) * L;          // rows of the last share
    static_assert(LAST >= 1 && LAST <= L && (2 * ROWS <= 256 || 2 * L <= 85), "two shares of doubled products between reliefs must fit 64 bits");
    u32 b0 = a_lds[0];
    u32 bn = a_lds[bstride];
    mont_lanes_sqr_row_asm_first<L, LPE>(T, a, b0, b0 << 1, n, n0inv, lowmask, nottopmask);
#pragma unroll 1
    for (int u = 0; u < LPE - (LAST < L ? 1 : 0); ++u) {
        if constexpr (2 * ROWS > 256) {                 // doubled products: at most 85 rows between reliefs = two shares of L <= 42
            if (u > 0 && u % 2 == 0) relieve_columns<L, LPE>(T, lowmask, nottopmask);
        }
        mont_sqr_lanes_share<L, LPE, ROWS, L, 0>(T, a, a_lds, bstride, n, n0inv, lowmask, nottopmask, bn, u * L, u == 0 ? 1 : 0);
    }
    if constexpr (LAST < L)
        mont_sqr_lanes_share<L, LPE, ROWS, LAST, 0>(T, a, a_lds, bstride, n, n0inv, lowmask, nottopmask, bn, (LPE - 1) * L, 0);
}

// Resolve the lazy columns into 28-bit limbs (value unchanged, < 2N < 2^(28*S)).
template <int S>
__device__ __forceinline__ void normalize_columns(u32 (&out)[S], const u64 (&T)[S]) {
    u64 c = 0;
#pragma unroll
    for (int j = 0; j < S; ++j) {
        c += T[j];
        out[j] = (u32)c & LIMB_MASK;
        c >>= LIMB_BITS;
    }
}

// x (limbs, value < 2N) -> canonical x mod N (< N), branch-free.
template <int S>
__device__ __forceinline__ void canonicalize(u32 (&x)[S], const u32* __restrict__ nmod) {
    u32 d[S];
    int32_t borrow = 0;
#pragma unroll
    for (int j = 0; j < S; ++j) {
        int32_t v = (int32_t)x[j] - (int32_t)nmod[j] + borrow;   // limbs < 2^28: no int32 overflow
        d[j] = (u32)v & LIMB_MASK;
        borrow = v >> LIMB_BITS;                                  // 0 or -1
    }
    bool ge = (borrow == 0);                                      // x >= N
#pragma unroll
    for (int j = 0; j < S; ++j) x[j] = ge ? d[j] : x[j];
}

}  // namespace vmn
