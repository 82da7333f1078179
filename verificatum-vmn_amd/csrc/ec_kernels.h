// ec_kernels.h — elliptic-curve groups (ECqPGroup: NIST P-224 / P-256 / P-384 / P-521, a = -3) on gfx950.
//
// The reference's code is group-agnostic (every call site of K1-K7 is reached with
// `pGroup` = ECqPGroup, the default group being P-256, demo/mixnet/.conf:153).  In VCR's multiplicative
// notation "mul" is point addition and "exp" scalar multiplication; these kernels mirror the modular
// ones family by family.
//
// One point per lane, everything in registers (a field element is S = 10 limbs of 28 bits for P-256,
// 14 for P-384, in Montgomery form mod p): a field product is a fully unrolled CIOS of 2*S^2
// v_mad_u64_u32 with lazily reduced operands — no LDS, no carries inside the product (same column
// argument as mont28.h).  Field values are kept "lazy": limbs normalised to 28 bits, value allowed to be
// a small multiple of p (products come out < 2p -- < 17p inside the point formulas of P-256, mont_row -- sums add their
// bounds, a difference adds 64p);
// Montgomery products accept operands up to 2^12 p, so no reduction is needed inside a point operation.
//
// Device row of a point: Jacobian (X, Y, Z), 3*FW words (FW = S rounded up to 4), the last padding word
// holds the infinity flag.  Jacobian storage means no inversion anywhere except export.  Exceptional
// cases of the addition (equal points, opposite points, infinity) are detected by canonical zero tests
// and handled exactly (rare wave-divergent path), so every result is the true group element.
#pragma once
#include "modp_kernels.h"

namespace vmn {

// Curve constants in device memory (wave-uniform: read through scalar loads)
struct ECDev {
    const u32* p;      // S limbs of the field prime
    const u32* one;    // R mod p                  (Montgomery one)
    const u32* rr;     // R^2 mod p                (to Montgomery form)
    const u32* b;      // curve coefficient b, Montgomery form
    const u32* mp;     // 64 * p, normalised limbs (added before a subtraction so the result stays positive)
    const u32* mp2;    // 256 * p, for the few subtractions whose subtrahend is itself a difference (< 256 p)
    const u32* pm2;    // p - 2 as packed 32-bit words (Fermat inversion)
    const u32* pp14;   // (p + 1) / 4 as packed words: square roots when p = 3 mod 4 (P-256, P-384, P-521)
    // p = 1 mod 4 (P-224: p - 1 = 2^96 (2^128 - 1)): Tonelli-Shanks.  ts_s = 0 selects the (p + 1) / 4 power above.
    int ts_s;          // p - 1 = 2^ts_s Q, Q odd
    int ts_ewords;     // words of ts_e
    const u32* ts_e;   // (Q - 1) / 2 as packed words
    const u32* ts_c;   // z^Q for a fixed non-residue z, Montgomery form, S limbs
    u32 n0inv;         // -p^{-1} mod 2^28
    u32 p1p;           // p[1] + 1 (limb 1 of the prime, plus the carry fold of mont_row)
    u32 c16;           // 16, as a run-time value: `hi * 16 + c` must stay ONE v_mad_u64_u32 (mont_row, wide digits)
    int pwords;        // words of pm2
};

template <int S>
struct ECfg {
    static constexpr int FS = S;
    static constexpr int FW = stride_for_limbs(S);
    static constexpr int ROW = 3 * FW;             // words per point row
    static constexpr int FLAG = ROW - 1;           // infinity flag word
    static_assert(FW > S, "a point row keeps its infinity flag in the padding word behind Z: the limb count must not be a multiple of 4");
    // waves per SIMD the point kernels are compiled for (VGPR budget 512 / MINW): measured, see DESIGN.md §5
#ifndef VMN_EC_MINW
#define VMN_EC_MINW 2
#endif
    static constexpr int MINW = S <= 10 ? VMN_EC_MINW : 1;
    // The two kernels that add runs of normalised rows into XYZZ registers (first bucket level, fixed-base powers) fit 168
    // registers without a spill (135 / 125 used when asked; tools/resource_usage.sh) -- the others do not.
#ifndef VMN_EC_MINW_RUN
#define VMN_EC_MINW_RUN 3
#endif
    static constexpr int MINW_RUN = S <= 10 ? VMN_EC_MINW_RUN : 1;
};

// ---------------------------------------------------------------------------------------------
// field arithmetic (lazy values, normalised limbs)
// ---------------------------------------------------------------------------------------------
template <int S>
__device__ __forceinline__ void f_norm(u32 (&r)[S], const u64 (&v)[S]) {
    u64 c = 0;
#pragma unroll
    for (int j = 0; j < S; ++j) {
        c += v[j];
        r[j] = (u32)c & LIMB_MASK;
        c >>= LIMB_BITS;
    }
}
// The field primes of the two curves are known at compile time (S = 10 <=> P-256, S = 15 <=> P-384; the host refuses any
// other prime of these sizes, vmn_ec_group_create): -p^-1 mod 2^28 is 1 for both, so the Montgomery quotient digit is the
// low limb itself, and a reduction row skips the zero limbs of p -- three of P-256's ten (p = 2^256 - 2^224 + 2^192 +
// 2^96 - 1 in radix 2^28 = fffffff fffffff fffffff 0000fff 0 0 1000000 0 fffffff 000000f), two of P-384's fifteen:
// 7 multiply-adds per row instead of 10 and no multiplication for m.
template <int S>
struct FieldPrime {
    static constexpr bool known = false;
};
template <>
struct FieldPrime<10> {
    static constexpr bool known = true;
    static constexpr u32 limb[10] = {0xfffffffu, 0xfffffffu, 0xfffffffu, 0x0000fffu, 0u, 0u, 0x1000000u, 0u, 0xfffffffu, 0xfu};
};
template <>
struct FieldPrime<15> {
    static constexpr bool known = true;
    static constexpr u32 limb[15] = {0xfffffffu, 0xfu, 0u, 0xffff000u, 0xffeffffu, 0xfffffffu, 0xfffffffu, 0xfffffffu, 0xfffffffu,
                                     0xfffffffu, 0xfffffffu, 0xfffffffu, 0xfffffffu, 0x00fffffu, 0u};
};
// one reduction row of a CIOS product: P <- (P + m p) / 2^28 with m = -P[0] / p mod 2^28; the top column is left to the caller
// WIDE rows (the point formulas over P-256): the quotient digit is the whole low WORD of column 0, x = P[0] mod 2^32, not its
// low 28 bits -- any digit = -P[0] / p mod 2^28 clears the limb, and x is one (x p = -x mod 2^28).  With p[0] = 2^28 - 1,
// P[0] + x p[0] = (P[0] - x) + x 2^28 = hi 2^32 + x 2^28 (hi = P[0] >> 32): its carry into column 1 is 16 hi + x, so the row is
//     P[0] <- x (p[1] + 1) + (16 hi + P[1]),    P[j-1] <- x p[j] + P[j]
// -- two multiply-adds and NOTHING else where the 28-bit digit pays a mask, a 64-bit shift and a 64-bit add: 20 instructions
// fewer per product of ten limbs (of ~216).  The price: a digit below 2^32 instead of 2^28 adds up to 16 p to the result, so a
// product leaves below 17 p, not 2 p (the bounds of the point formulas are stated for that: pt_dbl, pt_add, pt_madd), and a
// column holds S operand products (< 2^58: one operand may be a carry-less sum) + NZ digit products (< 2^60, NZ = non-zero
// limbs of p above limb 0) -- wide_digit_ok checks that this stays below 2^64: P-256 (6 of 9) yes, P-384 (13 of 14) no.
// Everything that canonicalises, compares, inverts or exports keeps the exact rows (results < 2 p).
template <int S>
constexpr bool wide_digit_ok() {
    using FP = FieldPrime<S>;
    if constexpr (!FP::known) {
        return false;
    } else {
        if (FP::limb[0] != LIMB_MASK || FP::limb[1] == 0) return false;
        unsigned nz = 0;
        for (int j = 1; j < S; ++j) nz += FP::limb[j] != 0;
        // in units of 2^58: S + 4 nz, + 1 for the carries (16 hi < 2^36) and the column shifted in; 2^64 = 64 units
        return (unsigned)S + 4u * nz + 1u < 64u;
    }
}
template <int S, bool WIDE = false>
__device__ __forceinline__ void mont_row(u64 (&P)[S], const ECDev& E) {
    using FP = FieldPrime<S>;
    if constexpr (WIDE) {
        static_assert(wide_digit_ok<S>(), "wide quotient digits need p = -1 mod 2^28 and room in the columns");
        const u32 x = (u32)P[0], hi = (u32)(P[0] >> 32);
        P[0] = (u64)x * E.p1p + ((u64)hi * E.c16 + P[1]);  // (a literal 16 becomes shift + mask + 64-bit add)
#pragma unroll
        for (int j = 2; j < S; ++j) {
            if (FP::limb[j] == 0) P[j - 1] = P[j];
            else P[j - 1] = (u64)x * E.p[j] + P[j];
        }
    } else if constexpr (FP::known) {
        // The VALUES of the non-zero limbs still come from E.p (scalar registers), not from the table: with literal
        // constants the compiler "strength-reduces" m * 2^24 and m * 15 into shifts / 32-bit multiplies plus 64-bit adds --
        // two or three issue slots where one v_mad_u64_u32 does it (seen in the ISA; every VALU instruction costs the
        // same slot on this machine).  Only WHICH limbs are zero, and n0inv = 1, are compile-time knowledge.
        // Both primes end in p[0] = 2^28 - 1, so the carry out of column 0 needs no product: P[0] + m (2^28 - 1) has the low
        // limb cleared by construction and its upper part is (P[0] >> 28) + m; folded into column 1 that is
        // m (p[1] + 1) + P[1] + (P[0] >> 28) -- one multiply-add (with p[1] + 1 from E.p1p), one shift, one 64-bit add.
        static_assert(FP::limb[0] == LIMB_MASK && FP::limb[1] != 0, "the carry fold assumes p = -1 mod 2^28");
        const u32 m = (u32)P[0] & LIMB_MASK;                           // n0inv = 1
        const u64 up = P[0] >> LIMB_BITS;
        P[0] = (u64)m * E.p1p + (P[1] + up);
#pragma unroll
        for (int j = 2; j < S; ++j) {
            if (FP::limb[j] == 0) P[j - 1] = P[j];                     // (the column simply gets no product in this row)
            else P[j - 1] = (u64)m * E.p[j] + P[j];
        }
    } else {
        const u32 m = ((u32)P[0] * E.n0inv) & LIMB_MASK;
        const u64 c = ((u64)m * E.p[0] + P[0]) >> LIMB_BITS;
#pragma unroll
        for (int j = 1; j < S; ++j) P[j - 1] = (u64)m * E.p[j] + P[j];
        P[0] += c;
    }
}
// r = a * b / R  (mod p), result < 2p for operands with a*b < R*p
template <int S, bool WIDE = false>
__device__ __forceinline__ void f_mul(u32 (&r)[S], const u32 (&a)[S], const u32 (&b)[S], const ECDev& E) {
    u64 P[S];
#pragma unroll
    for (int i = 0; i < S; ++i) {
#pragma unroll
        for (int j = 0; j < S; ++j) {
            if (i == 0 || j == S - 1) P[j] = (u64)a[j] * b[i];
            else P[j] = (u64)a[j] * b[i] + P[j];
        }
        mont_row<S, WIDE>(P, E);
    }
    P[S - 1] = 0;
    f_norm<S>(r, P);
}
// r = a^2 / R (mod p): row i multiplies only the limbs j >= i, the later ones by 2 a[i] -- each cross product a[i] a[j] is
// formed once, doubled, in row min(i, j) (an earlier row than in f_mul, the same column; a column is complete when it is
// reduced, in row i + j >= max(i, j)).  S (S + 1) / 2 + S^2 multiply-adds instead of 2 S^2.  Column bound: the operand may be
// a carry-less sum (f_addl: limbs < 2^29), so a column holds at most S products < 2^58 (a doubled cross product counts twice)
// + S reduction products < 2^56: < 2^62.8 for S <= 21.
template <int S, bool WIDE = false>
__device__ __forceinline__ void f_sqr(u32 (&r)[S], const u32 (&a)[S], const ECDev& E) {
    u64 P[S];
#pragma unroll
    for (int i = 0; i < S; ++i) {
        const u32 ai = a[i], ai2 = a[i] << 1;
#pragma unroll
        for (int j = i; j < S; ++j) {                                  // columns below i keep their running sums
            const u32 mult = j == i ? ai : ai2;
            if (i == 0 || j == S - 1) P[j] = (u64)a[j] * mult;         // a fresh column (the top one is vacated by every shift)
            else P[j] = (u64)a[j] * mult + P[j];
        }
        mont_row<S, WIDE>(P, E);
    }
    P[S - 1] = 0;
    f_norm<S>(r, P);
}

template <int S>
__device__ __forceinline__ void f_add(u32 (&r)[S], const u32 (&a)[S], const u32 (&b)[S]) {
    u32 c = 0;
#pragma unroll
    for (int j = 0; j < S; ++j) {
        c += a[j] + b[j];
        r[j] = j == S - 1 ? c : (c & LIMB_MASK);      // the top limb keeps the excess (value stays < 2^(28 S))
        c = j == S - 1 ? 0 : (c >> LIMB_BITS);
    }
}
// r = a + b limb by limb, NO carries: the limbs of the sum reach 2^29 (2^29.6 for a sum of three).  Only for sums that are
// consumed at once by a product (a column of f_mul / f_sqr holds S products of limbs: 10 x 2^(29 + 29) or 15 x 2^58 < 2^62
// leaves room for the reduction rows), or as the SUBTRAHEND / minuend of f_sub (a signed carry chain in 32 bits: limbs below
// 2^30 cannot overflow it).  Never stored, never compared, never the operand of f_small: one instruction per limb instead of three.
template <int S>
__device__ __forceinline__ void f_addl(u32 (&r)[S], const u32 (&a)[S], const u32 (&b)[S]) {
#pragma unroll
    for (int j = 0; j < S; ++j) r[j] = a[j] + b[j];
}
// A necessary condition for a = 0 mod p that costs four instructions, for a value with NORMALISED limbs (the result of
// f_sub) below 2^28 p: then a = k p with k < 2^28, and for a prime that is -1 modulo 2^84 (limbs 0..2 all ones: P-256) the
// limbs 1 and 2 of k p = k 2^84 (...) - k are all ones (k > 0) or zero (k = 0).  A random difference passes with probability
// 2^-55; only then does the caller pay the canonical test (f_is_zero: the reduction half of a product).  Other primes: true.
template <int S>
__device__ __forceinline__ bool f_maybe_zero(const u32 (&a)[S]) {
    using FP = FieldPrime<S>;
    if constexpr (FP::known) {
        if constexpr (FP::limb[0] == LIMB_MASK && FP::limb[1] == LIMB_MASK && FP::limb[2] == LIMB_MASK)
            return (a[1] & a[2]) == LIMB_MASK || (a[1] | a[2]) == 0;
    }
    return true;
}
// r = a - b + 64p  (b must be < 64p);  BIG: r = a - b + 256p (b < 256p).
// Bounds inside the point formulas (multiples of p; c = 2 for exact products, 17 for the wide-digit products of P-256,
// mont_row): products < c, sums of up to three of them (every subtrahend of the 64p form) < 3c = 51 < 64, a coordinate
// as it leaves a formula < c + 64 = 81, the differences with a coordinate as subtrahend (and the negation) use BIG and stay
// < c + 256 = 273, a doubled one < 546, Z of the all-affine addition (2H) < 546; the largest operand pair is
// (Z1 + Z2)^2 < 1200^2 = 2^20.5 -- every product has operands whose bounds multiply to less than 2^24 (the Montgomery limit
// R / p).  pt_dbl multiplies two products by 8 before subtracting them in the 64p form: those two stay exact (< 2).
template <int S, bool BIG = false>
__device__ __forceinline__ void f_sub(u32 (&r)[S], const u32 (&a)[S], const u32 (&b)[S], const ECDev& E) {
    const u32* __restrict__ mp = BIG ? E.mp2 : E.mp;
    int32_t c = 0;
#pragma unroll
    for (int j = 0; j < S; ++j) {
        int32_t v = (int32_t)a[j] + (int32_t)mp[j] - (int32_t)b[j] + c;
        if (j == S - 1) {
            r[j] = (u32)v;
        } else {
            r[j] = (u32)v & LIMB_MASK;
            c = v >> LIMB_BITS;
        }
    }
}
// r = -a + 64p  (a < 64p)
template <int S>
__device__ __forceinline__ void f_neg(u32 (&r)[S], const u32 (&a)[S], const ECDev& E) {
    int32_t c = 0;
#pragma unroll
    for (int j = 0; j < S; ++j) {
        int32_t v = (int32_t)E.mp[j] - (int32_t)a[j] + c;
        if (j == S - 1) {
            r[j] = (u32)v;
        } else {
            r[j] = (u32)v & LIMB_MASK;
            c = v >> LIMB_BITS;
        }
    }
}
// r = k * a for a small constant k (k * limb < 2^32)
template <int S, int K>
__device__ __forceinline__ void f_small(u32 (&r)[S], const u32 (&a)[S]) {
    u32 c = 0;
#pragma unroll
    for (int j = 0; j < S; ++j) {
        u32 v = a[j] * K + c;
        r[j] = j == S - 1 ? v : (v & LIMB_MASK);
        c = v >> LIMB_BITS;
    }
}
// canonical representative (< p) of a lazy value
template <int S>
__device__ __forceinline__ void f_canon(u32 (&r)[S], const u32 (&a)[S], const ECDev& E) {
    u32 one[S], t[S], d[S];
#pragma unroll
    for (int j = 0; j < S; ++j) one[j] = E.one[j];
    f_mul<S>(t, a, one, E);                            // a * R / R = a, now < 2p
    int32_t borrow = 0;
#pragma unroll
    for (int j = 0; j < S; ++j) {
        int32_t v = (int32_t)t[j] - (int32_t)E.p[j] + borrow;
        d[j] = (u32)v & LIMB_MASK;
        borrow = v >> LIMB_BITS;
    }
#pragma unroll
    for (int j = 0; j < S; ++j) r[j] = borrow == 0 ? d[j] : t[j];
}
// a = 0 mod p ?  Only the reduction half of a product (a * 1 / R): S^2 multiply-adds.  The result is = a / R mod p and
// lies in [0, p] (a / R < 1 for every lazy value), so a = 0 mod p exactly when it is 0 or p.
template <int S>
__device__ __forceinline__ bool f_is_zero(const u32 (&a)[S], const ECDev& E) {
    u64 P[S];
#pragma unroll
    for (int j = 0; j < S; ++j) P[j] = a[j];
#pragma unroll
    for (int i = 0; i < S; ++i) {
        mont_row<S>(P, E);
        P[S - 1] = 0;
    }
    u32 t[S];
    f_norm<S>(t, P);
    u32 nz = 0, np = 0;
#pragma unroll
    for (int j = 0; j < S; ++j) {
        nz |= t[j];
        np |= t[j] ^ E.p[j];
    }
    return nz == 0 || np == 0;
}
// r = a^(p-2): Fermat inversion, left-to-right binary (uniform exponent: no divergence).  Export only.
template <int S>
__device__ void f_pow_words(u32 (&r)[S], const u32 (&a)[S], const u32* __restrict__ ewords, int nwords, const ECDev& E) {
    u32 acc[S];
#pragma unroll
    for (int j = 0; j < S; ++j) acc[j] = E.one[j];
    for (int bit = nwords * 32 - 1; bit >= 0; --bit) {
        f_sqr<S>(acc, acc, E);
        if ((ewords[bit >> 5] >> (bit & 31)) & 1) f_mul<S>(acc, acc, a, E);
    }
#pragma unroll
    for (int j = 0; j < S; ++j) r[j] = acc[j];
}
template <int S>
__device__ void f_inv(u32 (&r)[S], const u32 (&a)[S], const ECDev& E) {
    f_pow_words<S>(r, a, E.pm2, E.pwords, E);
}

// z = a candidate square root of a (canonical Montgomery-form input): z^2 = a when a is a square; the caller checks.
// p = 3 mod 4: z = a^((p+1)/4).  p = 1 mod 4: Tonelli-Shanks with p - 1 = 2^s Q -- x = a^((Q+1)/2), t = a^Q, then while
// t != 1: i = the least exponent with t^(2^i) = 1, b = c^(2^(M-i-1)), x *= b, c = b^2, t *= c, M = i (c starts as z^Q for a
// non-residue z).  The loops are data-dependent (lanes of a wave take the longest of their paths): only the derivation of
// random points runs this, once per candidate.
template <int S>
__device__ void f_sqrt(u32 (&z)[S], const u32 (&a)[S], const ECDev& E) {
    if (E.ts_s == 0) {
        f_pow_words<S>(z, a, E.pp14, E.pwords, E);
        return;
    }
    u32 one[S], u[S], x[S], t[S], c[S], b[S], tt[S], d[S];
#pragma unroll
    for (int j = 0; j < S; ++j) {
        one[j] = E.one[j];
        c[j] = E.ts_c[j];
    }
    auto is_one = [&](const u32 (&v)[S]) {
        f_sub<S>(d, v, one, E);
        return f_is_zero<S>(d, E);
    };
    f_pow_words<S>(u, a, E.ts_e, E.ts_ewords, E);          // a^((Q-1)/2)
    f_mul<S>(x, a, u, E);                                  // a^((Q+1)/2)
    f_mul<S>(t, x, u, E);                                  // a^Q
    f_canon<S>(t, t, E);
    int M = E.ts_s;
    for (int guard = 0; guard <= E.ts_s; ++guard) {        // M strictly decreases: at most ts_s rounds
        if (is_one(t) || f_is_zero<S>(t, E)) break;        // done (t = 0: a = 0, x = 0 is its root)
        int i = 0;
#pragma unroll
        for (int j = 0; j < S; ++j) tt[j] = t[j];
        do {
            f_sqr<S>(tt, tt, E);
            f_canon<S>(tt, tt, E);
            ++i;
        } while (i < M && !is_one(tt));
        if (i >= M) break;                                 // a is not a square: the caller's check rejects x
#pragma unroll
        for (int j = 0; j < S; ++j) b[j] = c[j];
        for (int k = 0; k < M - i - 1; ++k) {
            f_sqr<S>(b, b, E);
            f_canon<S>(b, b, E);
        }
        f_mul<S>(x, x, b, E);
        f_sqr<S>(c, b, E);
        f_canon<S>(c, c, E);
        f_mul<S>(t, t, c, E);
        f_canon<S>(t, t, E);
        M = i;
    }
    f_canon<S>(z, x, E);
}

// ---------------------------------------------------------------------------------------------
// points
// ---------------------------------------------------------------------------------------------
template <int S>
struct Pt {
    u32 X[S], Y[S], Z[S];
    u32 inf;
};

template <int S>
__device__ __forceinline__ void pt_set_inf(Pt<S>& P, const ECDev& E) {
#pragma unroll
    for (int j = 0; j < S; ++j) {
        P.X[j] = E.one[j];
        P.Y[j] = E.one[j];
        P.Z[j] = 0;
    }
    P.inf = 1;
}
template <int S>
__device__ __forceinline__ void f_load(u32 (&a)[S], const u32* __restrict__ p) {
    constexpr int FW = stride_for_limbs(S);
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int k = 0; k < FW / 4; ++k) {
        uint4 v = q[k];
        if (4 * k + 0 < S) a[4 * k + 0] = v.x;
        if (4 * k + 1 < S) a[4 * k + 1] = v.y;
        if (4 * k + 2 < S) a[4 * k + 2] = v.z;
        if (4 * k + 3 < S) a[4 * k + 3] = v.w;
    }
}
template <int S>
__device__ __forceinline__ void pt_load(Pt<S>& P, const u32* __restrict__ row) {
    constexpr int FW = stride_for_limbs(S);
    f_load<S>(P.X, row);
    f_load<S>(P.Y, row + FW);
    f_load<S>(P.Z, row + 2 * FW);
    P.inf = row[3 * FW - 1];
}
// a NORMALISED row as the second operand of pt_madd: X, Y and the flag -- Z (= one) is not read
template <int S>
__device__ __forceinline__ void pt_load_normalised(Pt<S>& P, const u32* __restrict__ row) {
    constexpr int FW = stride_for_limbs(S);
    f_load<S>(P.X, row);
    f_load<S>(P.Y, row + FW);
    P.inf = row[3 * FW - 1];
}
template <int S>
__device__ __forceinline__ void pt_store(u32* __restrict__ row, const Pt<S>& P) {
    constexpr int FW = stride_for_limbs(S);
    auto put = [&](u32* dst, const u32 (&a)[S], u32 last) {
        uint4* q = reinterpret_cast<uint4*>(dst);
#pragma unroll
        for (int k = 0; k < FW / 4; ++k) {
            uint4 v;
            v.x = 4 * k + 0 < S ? a[4 * k + 0] : 0;
            v.y = 4 * k + 1 < S ? a[4 * k + 1] : 0;
            v.z = 4 * k + 2 < S ? a[4 * k + 2] : 0;
            v.w = 4 * k + 3 < S ? a[4 * k + 3] : (4 * k + 3 == FW - 1 ? last : 0);
            q[k] = v;
        }
    };
    put(row, P.X, 0);
    put(row + FW, P.Y, 0);
    put(row + 2 * FW, P.Z, P.inf);
}

// dbl-2001-b (a = -3): 3M + 5S, valid for every input (infinity stays infinity through the flag)
template <int S>
__device__ __forceinline__ void pt_dbl(Pt<S>& R, const Pt<S>& P, const ECDev& E) {
    constexpr bool W = wide_digit_ok<S>();             // products below 17 p instead of 2 p (mont_row)
    u32 delta[S], gamma[S], beta[S], alpha[S], t1[S], t2[S], t3[S];
    f_sqr<S, W>(delta, P.Z, E);
    f_sqr<S, W>(gamma, P.Y, E);
    f_mul<S, false>(beta, P.X, gamma, E);
    f_sub<S>(t1, P.X, delta, E);
    f_addl<S>(t2, P.X, delta);
    f_mul<S, W>(t3, t1, t2, E);
    f_small<S, 3>(alpha, t3);                          // alpha = 3 (X - delta)(X + delta)
    f_addl<S>(t1, P.Y, P.Z);
    f_sqr<S, W>(t2, t1, E);
    f_addl<S>(t3, gamma, delta);
    u32 Z3[S];
    f_sub<S>(Z3, t2, t3, E);                           // (Y + Z)^2 - gamma - delta
    f_sqr<S, W>(t1, alpha, E);
    f_small<S, 8>(t2, beta);
    u32 X3[S];
    f_sub<S>(X3, t1, t2, E);                           // alpha^2 - 8 beta
    f_small<S, 4>(t1, beta);
    f_sub<S, true>(t2, t1, X3, E);
    f_mul<S, W>(t3, alpha, t2, E);
    f_sqr<S, false>(t1, gamma, E);
    f_small<S, 8>(t2, t1);
    f_sub<S>(R.Y, t3, t2, E);                          // alpha (4 beta - X3) - 8 gamma^2
#pragma unroll
    for (int j = 0; j < S; ++j) {
        R.X[j] = X3[j];
        R.Z[j] = Z3[j];
    }
    R.inf = P.inf;
}

// add-2007-bl: 11M + 5S; exceptional inputs handled exactly
template <int S>
__device__ __forceinline__ void pt_add(Pt<S>& R, const Pt<S>& P, const Pt<S>& Q, const ECDev& E) {
    constexpr bool W = wide_digit_ok<S>();             // products below 17 p instead of 2 p (mont_row)
    u32 Z1Z1[S], Z2Z2[S], U1[S], U2[S], S1[S], S2[S], H[S], rr[S], t1[S], t2[S];
    f_sqr<S, W>(Z1Z1, P.Z, E);
    f_sqr<S, W>(Z2Z2, Q.Z, E);
    f_mul<S, W>(U1, P.X, Z2Z2, E);
    f_mul<S, W>(U2, Q.X, Z1Z1, E);
    f_mul<S, W>(t1, P.Y, Q.Z, E);
    f_mul<S, W>(S1, t1, Z2Z2, E);
    f_mul<S, W>(t1, Q.Y, P.Z, E);
    f_mul<S, W>(S2, t1, Z1Z1, E);
    f_sub<S>(H, U2, U1, E);
    f_sub<S>(rr, S2, S1, E);
    bool hz = f_maybe_zero<S>(H) && f_is_zero<S>(H, E);
    bool special = P.inf || Q.inf || hz;
    Pt<S> G;                                           // general-case result
    {
        u32 I[S], J[S], r[S], V[S];
        f_addl<S>(t1, H, H);
        f_sqr<S, W>(I, t1, E);                            // (2H)^2
        f_mul<S, W>(J, H, I, E);
        f_addl<S>(r, rr, rr);
        f_mul<S, W>(V, U1, I, E);
        f_sqr<S, W>(t1, r, E);
        f_addl<S>(t2, V, V);
        f_addl<S>(t2, t2, J);
        f_sub<S>(G.X, t1, t2, E);                      // r^2 - J - 2V
        f_sub<S, true>(t1, V, G.X, E);
        f_mul<S, W>(t2, r, t1, E);
        f_mul<S, W>(t1, S1, J, E);
        f_addl<S>(t1, t1, t1);
        f_sub<S>(G.Y, t2, t1, E);                      // r (V - X3) - 2 S1 J
        f_addl<S>(t1, P.Z, Q.Z);
        f_sqr<S, W>(t2, t1, E);
        f_addl<S>(t1, Z1Z1, Z2Z2);
        f_sub<S>(t2, t2, t1, E);
        f_mul<S, W>(G.Z, t2, H, E);                       // ((Z1 + Z2)^2 - Z1Z1 - Z2Z2) H
        G.inf = 0;
    }
    if (special) {                                     // rare: wave-divergent
        if (P.inf) {
            G = Q;
        } else if (Q.inf) {
            G = P;
        } else if (f_is_zero<S>(rr, E)) {
            pt_dbl<S>(G, P, E);                        // P == Q
        } else {
            pt_set_inf<S>(G, E);                       // P == -Q
        }
    }
    R = G;
}

// madd-2007-bl: P (Jacobian) + Q with Q NORMALISED (Z = 1, or the infinity flag): 7M + 4S instead of 11M + 5S.  The rows of
// a normalised array keep the three-coordinate layout (Z = the Montgomery one), so every other kernel reads them as they
// are.  Bounds (multiples of p): X1, Y1, Z1 < 81 as they leave an addition or a doubling, so the differences with them as
// subtrahend use the 256p form (< 273); the largest product is r * (V - X3) < 546 * 273 p^2, far below R p = 2^24 p^2.
template <int S>
__device__ __forceinline__ void pt_madd(Pt<S>& R, const Pt<S>& P, const Pt<S>& Q, const ECDev& E) {
    constexpr bool W = wide_digit_ok<S>();             // products below 17 p instead of 2 p (mont_row)
    u32 Z1Z1[S], U2[S], S2[S], H[S], HH[S], I[S], J[S], r[S], V[S], t1[S], t2[S];
    f_sqr<S, W>(Z1Z1, P.Z, E);
    f_mul<S, W>(U2, Q.X, Z1Z1, E);
    f_mul<S, W>(t1, P.Z, Z1Z1, E);
    f_mul<S, W>(S2, Q.Y, t1, E);
    f_sub<S, true>(H, U2, P.X, E);
    f_sub<S, true>(t1, S2, P.Y, E);                    // S2 - Y1
    bool hz = f_maybe_zero<S>(H) && f_is_zero<S>(H, E);
    bool special = P.inf || Q.inf || hz;
    Pt<S> G;
    {
        f_sqr<S, W>(HH, H, E);
        f_small<S, 4>(I, HH);
        f_mul<S, W>(J, H, I, E);
        f_addl<S>(r, t1, t1);
        f_mul<S, W>(V, P.X, I, E);
        f_sqr<S, W>(t2, r, E);
        u32 t3[S];
        f_addl<S>(t3, V, V);
        f_addl<S>(t3, t3, J);
        f_sub<S>(G.X, t2, t3, E);                      // r^2 - J - 2V
        f_sub<S, true>(t2, V, G.X, E);
        f_mul<S, W>(t3, r, t2, E);
        f_mul<S, W>(t2, P.Y, J, E);
        f_addl<S>(t2, t2, t2);
        f_sub<S>(G.Y, t3, t2, E);                      // r (V - X3) - 2 Y1 J
        f_addl<S>(t2, P.Z, H);
        f_sqr<S, W>(t3, t2, E);
        f_addl<S>(t2, Z1Z1, HH);
        f_sub<S>(G.Z, t3, t2, E);                      // (Z1 + H)^2 - Z1Z1 - HH
        G.inf = 0;
    }
    if (special) {                                     // rare: wave-divergent
        if (P.inf) {
            G = Q;
#pragma unroll
            for (int j = 0; j < S; ++j) G.Z[j] = Q.inf ? 0u : E.one[j];       // (Q.Z is not loaded: it is one by contract)
        } else if (Q.inf) {
            G = P;
        } else if (f_is_zero<S>(t1, E)) {
            pt_dbl<S>(G, P, E);                        // P == Q
        } else {
            pt_set_inf<S>(G, E);                       // P == -Q
        }
    }
    R = G;
}

// mmadd-2007-bl: P + Q with BOTH operands normalised (Z = 1, or the infinity flag): 4M + 2S.  The first addition of every
// chunk of the first bucket-tree level adds two rows of the normalised input arrays (round 4: one addition in seven of that
// level at 6 field products instead of 11).  P.Z is not read.  Bounds: X, Y of a normalised row < 2 (a negated Y < 66), so
// H, Y2 - Y1 in the 256p form < 273 (wide-digit products: rows < 17); Z3 = 2H < 546, inside what the next mixed addition
// accepts (operands up to 2^12 p).
template <int S>
__device__ __forceinline__ void pt_mmadd(Pt<S>& R, const Pt<S>& P, const Pt<S>& Q, const ECDev& E) {
    constexpr bool W = wide_digit_ok<S>();
    u32 H[S], HH[S], I[S], J[S], r[S], V[S], t1[S], t2[S], t3[S];
    f_sub<S, true>(H, Q.X, P.X, E);
    f_sub<S, true>(t1, Q.Y, P.Y, E);                   // Y2 - Y1
    bool hz = f_maybe_zero<S>(H) && f_is_zero<S>(H, E);
    bool special = P.inf || Q.inf || hz;
    Pt<S> G;
    {
        f_sqr<S, W>(HH, H, E);
        f_small<S, 4>(I, HH);
        f_mul<S, W>(J, H, I, E);
        f_addl<S>(r, t1, t1);
        f_mul<S, W>(V, P.X, I, E);
        f_sqr<S, W>(t2, r, E);
        f_addl<S>(t3, V, V);
        f_addl<S>(t3, t3, J);
        f_sub<S>(G.X, t2, t3, E);                      // r^2 - J - 2V
        f_sub<S, true>(t2, V, G.X, E);
        f_mul<S, W>(t3, r, t2, E);
        f_mul<S, W>(t2, P.Y, J, E);
        f_addl<S>(t2, t2, t2);
        f_sub<S>(G.Y, t3, t2, E);                      // r (V - X3) - 2 Y1 J
        f_add<S>(G.Z, H, H);                           // 2 H
        G.inf = 0;
    }
    if (special) {                                     // rare: wave-divergent
        if (P.inf && Q.inf) {
            pt_set_inf<S>(G, E);
        } else if (P.inf || Q.inf) {
            const Pt<S>& T = P.inf ? Q : P;
#pragma unroll
            for (int j = 0; j < S; ++j) {
                G.X[j] = T.X[j];
                G.Y[j] = T.Y[j];
                G.Z[j] = E.one[j];                     // (Z of a normalised row is one by contract)
            }
            G.inf = 0;
        } else if (f_is_zero<S>(t1, E)) {
            Pt<S> D;                                   // P == Q: double the affine point
#pragma unroll
            for (int j = 0; j < S; ++j) {
                D.X[j] = P.X[j];
                D.Y[j] = P.Y[j];
                D.Z[j] = E.one[j];
            }
            D.inf = 0;
            pt_dbl<S>(G, D, E);
        } else {
            pt_set_inf<S>(G, E);                       // P == -Q
        }
    }
    R = G;
}

// ---------------------------------------------------------------------------------------------
// Running sums in XYZZ coordinates (x = X / ZZ, y = Y / ZZZ, ZZ^3 = ZZZ^2).  A lane that adds a run of NORMALISED rows into
// one sum -- a chunk of the first bucket level, the table rows of a fixed-base power -- pays Z1^2 and Z1^3 again in every
// mixed Jacobian addition (1S + 1M of its 7M + 4S).  With ZZ and ZZZ carried along the mixed addition is madd-2008-s,
// 8M + 2S and half the sums and differences; the sum goes back to a Jacobian row at the end of the run,
// (X ZZ, Y ZZZ, ZZ): 2M once (ZZ^2 = (ZZ)^2 and ZZ^3 = ZZZ^2, so the row means the same point).  Only registers change: rows
// in memory stay Jacobian.  Bounds as in pt_madd: X, Y < 81 (subtrahends of the 256p form), ZZ, ZZZ products.
// The exceptional cases go through the Jacobian code (pt_dbl) and come back: rare and wave-divergent.
// ---------------------------------------------------------------------------------------------
template <int S>
struct PtX {
    u32 X[S], Y[S], ZZ[S], ZZZ[S];
    u32 inf;
};
template <int S>
__device__ __forceinline__ void ptx_from_normalised(PtX<S>& A, const Pt<S>& P, const ECDev& E) {
#pragma unroll
    for (int j = 0; j < S; ++j) {
        A.X[j] = P.X[j];
        A.Y[j] = P.Y[j];
        A.ZZ[j] = E.one[j];
        A.ZZZ[j] = E.one[j];
    }
    A.inf = P.inf;
}
template <int S>
__device__ __forceinline__ void ptx_from_jacobian(PtX<S>& A, const Pt<S>& P, const ECDev& E) {
    f_sqr<S>(A.ZZ, P.Z, E);
    f_mul<S>(A.ZZZ, A.ZZ, P.Z, E);
#pragma unroll
    for (int j = 0; j < S; ++j) {
        A.X[j] = P.X[j];
        A.Y[j] = P.Y[j];
    }
    A.inf = P.inf;
}
template <int S>
__device__ __forceinline__ void ptx_to_jacobian(Pt<S>& R, const PtX<S>& A, const ECDev& E) {
    if (A.inf) {
        pt_set_inf<S>(R, E);
    } else {
        f_mul<S>(R.X, A.X, A.ZZ, E);
        f_mul<S>(R.Y, A.Y, A.ZZZ, E);
#pragma unroll
        for (int j = 0; j < S; ++j) R.Z[j] = A.ZZ[j];
        R.inf = 0;
    }
}
// the tail both additions share: from Pd = x2' - X1, Rd = y2' - Y1 (256p form), X1, Y1 to X3, Y3, PP, PPP
template <int S>
__device__ __forceinline__ void ptx_tail(u32 (&X3)[S], u32 (&Y3)[S], u32 (&PP)[S], u32 (&PPP)[S], const u32 (&Pd)[S], const u32 (&Rd)[S],
                                         const u32 (&X1)[S], const u32 (&Y1)[S], const ECDev& E) {
    constexpr bool W = wide_digit_ok<S>();
    u32 Qv[S], t1[S], t2[S], t3[S];
    f_sqr<S, W>(PP, Pd, E);
    f_mul<S, W>(PPP, Pd, PP, E);
    f_mul<S, W>(Qv, X1, PP, E);
    f_sqr<S, W>(t1, Rd, E);
    f_addl<S>(t2, Qv, Qv);
    f_addl<S>(t2, t2, PPP);                            // 2Q + PPP < 51
    f_sub<S>(X3, t1, t2, E);                           // R^2 - PPP - 2Q
    f_sub<S, true>(t1, Qv, X3, E);
    f_mul<S, W>(t2, Rd, t1, E);
    f_mul<S, W>(t3, Y1, PPP, E);
    f_sub<S>(Y3, t2, t3, E);                           // R (Q - X3) - Y1 PPP
}
// A += Q, Q normalised (madd-2008-s)
template <int S>
__device__ __forceinline__ void ptx_madd(PtX<S>& A, const Pt<S>& Q, const ECDev& E) {
    constexpr bool W = wide_digit_ok<S>();
    u32 U2[S], S2[S], Pd[S], Rd[S], PP[S], PPP[S], X3[S], Y3[S];
    f_mul<S, W>(U2, Q.X, A.ZZ, E);
    f_mul<S, W>(S2, Q.Y, A.ZZZ, E);
    f_sub<S, true>(Pd, U2, A.X, E);
    f_sub<S, true>(Rd, S2, A.Y, E);
    const bool hz = f_maybe_zero<S>(Pd) && f_is_zero<S>(Pd, E);
    const bool special = A.inf || Q.inf || hz;
    if (!special) {
        ptx_tail<S>(X3, Y3, PP, PPP, Pd, Rd, A.X, A.Y, E);
        f_mul<S, W>(A.ZZ, A.ZZ, PP, E);
        f_mul<S, W>(A.ZZZ, A.ZZZ, PPP, E);
#pragma unroll
        for (int j = 0; j < S; ++j) {
            A.X[j] = X3[j];
            A.Y[j] = Y3[j];
        }
    } else if (A.inf) {                                // rare from here on
        ptx_from_normalised<S>(A, Q, E);
    } else if (Q.inf) {
    } else if (f_is_zero<S>(Rd, E)) {                  // the same point: double the normalised one
        Pt<S> D, G;
#pragma unroll
        for (int j = 0; j < S; ++j) {
            D.X[j] = Q.X[j];
            D.Y[j] = Q.Y[j];
            D.Z[j] = E.one[j];
        }
        D.inf = 0;
        pt_dbl<S>(G, D, E);
        ptx_from_jacobian<S>(A, G, E);
    } else {
        A.inf = 1;                                     // opposite points
    }
}
// A = P + Q, both normalised (mmadd-2008-s: ZZ3 = PP, ZZZ3 = PPP)
template <int S>
__device__ __forceinline__ void ptx_mmadd(PtX<S>& A, const Pt<S>& P, const Pt<S>& Q, const ECDev& E) {
    u32 Pd[S], Rd[S];
    f_sub<S, true>(Pd, Q.X, P.X, E);
    f_sub<S, true>(Rd, Q.Y, P.Y, E);
    const bool hz = f_maybe_zero<S>(Pd) && f_is_zero<S>(Pd, E);
    const bool special = P.inf || Q.inf || hz;
    if (!special) {
        ptx_tail<S>(A.X, A.Y, A.ZZ, A.ZZZ, Pd, Rd, P.X, P.Y, E);
        A.inf = 0;
    } else if (P.inf) {
        ptx_from_normalised<S>(A, Q, E);
    } else if (Q.inf) {
        ptx_from_normalised<S>(A, P, E);
    } else if (f_is_zero<S>(Rd, E)) {
        Pt<S> D, G;
#pragma unroll
        for (int j = 0; j < S; ++j) {
            D.X[j] = P.X[j];
            D.Y[j] = P.Y[j];
            D.Z[j] = E.one[j];
        }
        D.inf = 0;
        pt_dbl<S>(G, D, E);
        ptx_from_jacobian<S>(A, G, E);
    } else {
        ptx_from_normalised<S>(A, P, E);               // (any finite coordinates: the flag is what counts)
        A.inf = 1;
    }
}

// ---------------------------------------------------------------------------------------------
// kernels (one point per lane; no LDS)
// ---------------------------------------------------------------------------------------------
// big-endian x || y (nbytes each; all 0xff = infinity) -> rows.  flags |= 1: coordinate >= p or point not on the
// curve (replaced by the identity, the reference's "trivial value" convention).
template <int S, int NW>
__global__ void __launch_bounds__(BLOCK) k_ec_import(u32* __restrict__ out, const uint8_t* __restrict__ be, size_t nbytes,
                                                     size_t stride, int framed, size_t n, ECDev E, u32* __restrict__ flags) {
    // framed: every point is the byte tree node(leaf(x), leaf(y)) = 00 00000002 | 01 len x | 01 len y (the form VCR gives a
    // curve point; [NOT-IN-REF]: restated from the verifier specification); the coordinates are moved together first
    using C1 = Cfg<S, 1>;
    constexpr int ROW = ECfg<S>::ROW;
    size_t el = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (el >= n) return;
    const uint8_t* rec = be + el * stride;
    uint8_t packed[2 * (4 * NW + 4)];
    const uint8_t* src = rec;
    if (framed) {
        auto hdr_ok = [&](const uint8_t* h, uint8_t tag, size_t len) {
            return h[0] == tag && h[1] == (uint8_t)(len >> 24) && h[2] == (uint8_t)(len >> 16) && h[3] == (uint8_t)(len >> 8) && h[4] == (uint8_t)len;
        };
        if (!(hdr_ok(rec, 0, 2) && hdr_ok(rec + 5, 1, nbytes) && hdr_ok(rec + 10 + nbytes, 1, nbytes))) atomicOr(flags, 4u);
        if (nbytes <= sizeof(packed) / 2) {
            for (size_t i = 0; i < nbytes; ++i) {
                packed[i] = rec[10 + i];
                packed[nbytes + i] = rec[15 + nbytes + i];
            }
            src = packed;
        } else {
            atomicOr(flags, 4u);
        }
    }
    u32 allff = 0xff;
    for (size_t i = 0; i < 2 * nbytes; ++i) allff &= src[i];
    Pt<S> P;
    bool bad = false;
    if (allff == 0xff) {
        pt_set_inf<S>(P, E);
    } else {
        Lane<C1> ln(nullptr);
        u32 x[S], y[S], pp[S], d[S];
#pragma unroll
        for (int j = 0; j < S; ++j) pp[j] = E.p[j];
        limbs_from_be<C1, NW>(x, src, (long)nbytes, ln);
        limbs_from_be<C1, NW>(y, src + nbytes, (long)nbytes, ln);
        u32 extra = 0;
        for (long o = (long)nbytes - 4L * NW - 1; o >= 0; --o) extra |= src[o] | src[nbytes + o];
        bad = extra != 0 || borrow_sweep<S>(d, x, pp, 0) == 0 || borrow_sweep<S>(d, y, pp, 0) == 0;
        u32 rrc[S];
#pragma unroll
        for (int j = 0; j < S; ++j) rrc[j] = E.rr[j];
        f_mul<S>(P.X, x, rrc, E);
        f_mul<S>(P.Y, y, rrc, E);
#pragma unroll
        for (int j = 0; j < S; ++j) P.Z[j] = E.one[j];
        P.inf = 0;
        // on the curve?  y^2 == x^3 - 3x + b
        u32 lhs[S], t[S], x3[S], rhs[S], bb[S];
#pragma unroll
        for (int j = 0; j < S; ++j) bb[j] = E.b[j];
        f_sqr<S>(lhs, P.Y, E);
        f_sqr<S>(t, P.X, E);
        f_mul<S>(x3, t, P.X, E);
        f_small<S, 3>(t, P.X);
        f_add<S>(rhs, x3, bb);
        f_sub<S>(rhs, rhs, t, E);                       // < 4p + 64p
        f_sub<S, true>(t, lhs, rhs, E);                 // subtrahend up to 68p: the 256p form
        bad = bad || !f_is_zero<S>(t, E);
        if (bad) pt_set_inf<S>(P, E);
    }
    if (bad) atomicOr(flags, 1u);
    pt_store<S>(out + el * ROW, P);
}

template <int S, int NW>
__global__ void __launch_bounds__(BLOCK) k_ec_export(uint8_t* __restrict__ be, size_t nbytes, size_t stride, int framed,
                                                     const u32* __restrict__ in, size_t n, ECDev E) {
    constexpr int ROW = ECfg<S>::ROW;
    size_t el = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (el >= n) return;
    Pt<S> P;
    pt_load<S>(P, in + el * ROW);
    uint8_t* dst = be + el * stride;
    uint8_t* dsty = dst + nbytes;
    const bool rows_normalised = (framed & 2) != 0;    // bit 1: Z = 1 in every row (the host normalised a large array first)
    framed &= 1;
    if (framed) {                                      // node(leaf(x), leaf(y)), see k_ec_import
        auto hdr = [&](uint8_t* h, uint8_t tag, size_t len) {
            h[0] = tag;
            h[1] = (uint8_t)(len >> 24);
            h[2] = (uint8_t)(len >> 16);
            h[3] = (uint8_t)(len >> 8);
            h[4] = (uint8_t)len;
        };
        hdr(dst, 0, 2);
        hdr(dst + 5, 1, nbytes);
        hdr(dst + 10 + nbytes, 1, nbytes);
        dsty = dst + 15 + nbytes;
        dst += 10;
    }
    if (P.inf) {                                       // the point at infinity: both coordinates -1 (all 0xff)
        for (size_t i = 0; i < nbytes; ++i) dst[i] = dsty[i] = 0xff;
        return;
    }
    u32 xa[S], ya[S], one1[S], t[S];
    if (rows_normalised) {                             // (wave-uniform)
#pragma unroll
        for (int j = 0; j < S; ++j) {
            xa[j] = P.X[j];
            ya[j] = P.Y[j];
        }
    } else {                                           // one Fermat power per point: ~380 products where the rest of the export is 5
        u32 zi[S], zi2[S], zi3[S];
        f_inv<S>(zi, P.Z, E);
        f_sqr<S>(zi2, zi, E);
        f_mul<S>(zi3, zi2, zi, E);
        f_mul<S>(xa, P.X, zi2, E);
        f_mul<S>(ya, P.Y, zi3, E);
    }
#pragma unroll
    for (int j = 0; j < S; ++j) one1[j] = j == 0 ? 1u : 0u;
    // leave the Montgomery domain (multiply by 1) and canonicalise
    auto out_coord = [&](const u32 (&v)[S], uint8_t* d) {
        u32 s[S], c[S], w[NW];
        f_mul<S>(s, v, one1, E);                       // v / R : standard representative, < 2p
        int32_t borrow = 0;
#pragma unroll
        for (int j = 0; j < S; ++j) {
            int32_t q = (int32_t)s[j] - (int32_t)E.p[j] + borrow;
            c[j] = (u32)q & LIMB_MASK;
            borrow = q >> LIMB_BITS;
        }
#pragma unroll
        for (int j = 0; j < S; ++j) t[j] = borrow == 0 ? c[j] : s[j];
        limbs_to_words<S, NW>(w, t);
#pragma unroll
        for (int k = 0; k < NW; ++k) store_be_word(d, (long)nbytes, k, w[k]);
        for (long o = (long)nbytes - 4L * NW - 1; o >= 0; --o) d[o] = 0;
    };
    out_coord(xa, dst);
    out_coord(ya, dsty);
}

// K4: out[i] = x[i] + y[i]   (ystride = 0: one shared point)
template <int S>
__global__ void __launch_bounds__(BLOCK, ECfg<S>::MINW) k_ec_add(u32* __restrict__ out, const u32* __restrict__ x, const u32* __restrict__ y,
                                                  size_t ystride, size_t n, ECDev E) {
    constexpr int ROW = ECfg<S>::ROW;
    size_t el = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (el >= n) return;
    Pt<S> P, Q, R;
    pt_load<S>(P, x + el * ROW);
    pt_load<S>(Q, y + el * ystride);
    pt_add<S>(R, P, Q, E);
    pt_store<S>(out + el * ROW, R);
}

// inverse of every element: (X, -Y, Z)
template <int S>
__global__ void __launch_bounds__(BLOCK) k_ec_neg(u32* __restrict__ out, const u32* __restrict__ x, size_t n, ECDev E) {
    constexpr int ROW = ECfg<S>::ROW;
    size_t el = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (el >= n) return;
    Pt<S> P;
    pt_load<S>(P, x + el * ROW);
    u32 z[S], y[S];
#pragma unroll
    for (int j = 0; j < S; ++j) z[j] = 0;
    f_sub<S, true>(y, z, P.Y, E);
#pragma unroll
    for (int j = 0; j < S; ++j) P.Y[j] = y[j];
    pt_store<S>(out + el * ROW, P);
}

// K6: flags |= 1 where x[i] != y[i] as group elements (cross-multiplied Jacobian comparison)
template <int S>
__global__ void __launch_bounds__(BLOCK) k_ec_equal(const u32* __restrict__ x, const u32* __restrict__ y, size_t n, ECDev E,
                                                    u32* __restrict__ flags) {
    constexpr int ROW = ECfg<S>::ROW;
    size_t el = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (el >= n) return;
    Pt<S> P, Q;
    pt_load<S>(P, x + el * ROW);
    pt_load<S>(Q, y + el * ROW);
    bool eq;
    if (P.inf || Q.inf) {
        eq = P.inf && Q.inf;
    } else {
        u32 a[S], b[S], c[S], d[S], t[S];
        f_sqr<S>(a, P.Z, E);
        f_sqr<S>(b, Q.Z, E);
        f_mul<S>(c, P.X, b, E);
        f_mul<S>(d, Q.X, a, E);
        f_sub<S>(t, c, d, E);
        eq = f_is_zero<S>(t, E);
        f_mul<S>(c, a, P.Z, E);
        f_mul<S>(d, b, Q.Z, E);
        f_mul<S>(a, P.Y, d, E);
        f_mul<S>(b, Q.Y, c, E);
        f_sub<S>(t, a, b, E);
        eq = eq && f_is_zero<S>(t, E);
    }
    if (!eq) atomicOr(flags, 1u);
}

// K1a / K1b: out[i] = e[i] * x[i]  (fixed window, per-lane table of multiples in scratch)
template <int S>
__global__ void __launch_bounds__(BLOCK, ECfg<S>::MINW) k_ec_mulvar(u32* __restrict__ out, const u32* __restrict__ x, const u32* __restrict__ e,
                                                     int ewords, size_t estride, int ebits, int wbits, size_t n, ECDev E,
                                                     u32* __restrict__ tab) {
    constexpr int ROW = ECfg<S>::ROW;
    const size_t ntiles = (n + BLOCK - 1) / BLOCK;
    const int tsize = 1 << wbits;
    u32* mytab = tab + ((size_t)blockIdx.x * BLOCK + threadIdx.x) * (size_t)tsize * ROW;
    const int nwin = (ebits + wbits - 1) / wbits;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        size_t el = t * BLOCK + threadIdx.x;
        bool live = el < n;
        size_t ec = live ? el : n - 1;
        const u32* ep = e + ec * estride;
        Pt<S> P, A;
        pt_load<S>(P, x + ec * ROW);
        pt_set_inf<S>(A, E);
        pt_store<S>(mytab, A);
        pt_store<S>(mytab + ROW, P);
        A = P;
#pragma unroll 1
        for (int k = 2; k < tsize; ++k) {
            pt_add<S>(A, A, P, E);
            pt_store<S>(mytab + (size_t)k * ROW, A);
        }
        u32 d = exp_digit(ep, ewords, (nwin - 1) * wbits, wbits);
        pt_load<S>(A, mytab + (size_t)d * ROW);
#pragma unroll 1
        for (int wi = nwin - 2; wi >= 0; --wi) {
#pragma unroll 1
            for (int s = 0; s < wbits; ++s) pt_dbl<S>(A, A, E);
            d = exp_digit(ep, ewords, wi * wbits, wbits);
            Pt<S> T;
            pt_load<S>(T, mytab + (size_t)d * ROW);
            pt_add<S>(A, A, T, E);
        }
        if (live) pt_store<S>(out + el * ROW, A);
    }
}

// out[i] = e * x[i] + f[i] * y[i]  (one shared scalar e, per-lane scalars f): the two scalar multiplications of a verifier's
// check (B) as ONE chain of doublings (round 4) -- B_i^v (B_{i-1}^-1)^{k_E,i} in the reference's multiplicative notation, the
// caller hands in y = -B_shift (a negation is free on a curve).  Fixed windows of wbits bits for both scalars, two per-lane
// tables in scratch: max(ebits, fbits) doublings + ceil(ebits / w) + ceil(fbits / w) additions + 2 (2^w - 2) for the tables,
// against twice the doublings of two k_ec_mulvar launches.
template <int S>
__global__ void __launch_bounds__(BLOCK, ECfg<S>::MINW) k_ec_mulvar2(u32* __restrict__ out, const u32* __restrict__ x, const u32* __restrict__ e,
                                                      int ewords, int ebits, const u32* __restrict__ y, const u32* __restrict__ f,
                                                      int fwords, size_t fstride, int fbits, int wbits, size_t n, ECDev E,
                                                      u32* __restrict__ tab) {
    constexpr int ROW = ECfg<S>::ROW;
    const size_t ntiles = (n + BLOCK - 1) / BLOCK;
    const int tsize = 1 << wbits;
    u32* tx = tab + ((size_t)blockIdx.x * BLOCK + threadIdx.x) * (size_t)(2 * tsize) * ROW;
    u32* ty = tx + (size_t)tsize * ROW;
    const int bits = ebits > fbits ? ebits : fbits;
    const int nwin = (bits + wbits - 1) / wbits;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        size_t el = t * BLOCK + threadIdx.x;
        bool live = el < n;
        size_t ec = live ? el : n - 1;
        const u32* fp = f + ec * fstride;
        Pt<S> A;
#pragma unroll 1
        for (int which = 0; which < 2; ++which) {          // the two tables: k * x and k * y, k < 2^w (entry 0: infinity)
            u32* mytab = which ? ty : tx;
            Pt<S> P;
            pt_load<S>(P, (which ? y : x) + ec * ROW);
            pt_set_inf<S>(A, E);
            pt_store<S>(mytab, A);
            pt_store<S>(mytab + ROW, P);
            A = P;
#pragma unroll 1
            for (int k = 2; k < tsize; ++k) {
                pt_add<S>(A, A, P, E);
                pt_store<S>(mytab + (size_t)k * ROW, A);
            }
        }
        pt_set_inf<S>(A, E);
#pragma unroll 1
        for (int wi = nwin - 1; wi >= 0; --wi) {
#pragma unroll 1
            for (int s = 0; s < wbits; ++s) pt_dbl<S>(A, A, E);      // (infinity stays infinity: the first window costs nothing real)
            const u32 de = wi * wbits < ebits ? exp_digit(e, ewords, wi * wbits, wbits) : 0u;
            const u32 df = wi * wbits < fbits ? exp_digit(fp, fwords, wi * wbits, wbits) : 0u;
            Pt<S> T;
            if (de) {                                                // (wave-uniform: e is shared)
                pt_load<S>(T, tx + (size_t)de * ROW);
                pt_add<S>(A, A, T, E);
            }
            pt_load<S>(T, ty + (size_t)df * ROW);                    // entry 0 is the identity: no branch per lane
            pt_add<S>(A, A, T, E);
        }
        if (live) pt_store<S>(out + el * ROW, A);
    }
}

// sq[j] = 2^j * base, j < count: the doubling chain of a fixed-base table, one lane (count ~ 256-400 doublings)
// (launch bounds: without them the compiler budgets registers for 1024 threads per block and the point doubling spills --
// the chain then took 27 us per doubling instead of 5)
template <int S>
__global__ void __launch_bounds__(64) k_ec_chain(u32* __restrict__ sq, const u32* __restrict__ base, int count, ECDev E) {
    constexpr int ROW = ECfg<S>::ROW;
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    Pt<S> A;
    pt_load<S>(A, base);
    for (int j = 0; j < count; ++j) {
        pt_store<S>(sq + (size_t)j * ROW, A);
        pt_dbl<S>(A, A, E);
    }
}

// K2 table level l: T[k][2^l + r] = T[k][r] + T[k][2^l]
template <int S>
__global__ void __launch_bounds__(BLOCK, ECfg<S>::MINW) k_ec_fixed_level(u32* __restrict__ T, int w, int nwin, int l, ECDev E) {
    constexpr int ROW = ECfg<S>::ROW;
    size_t per = ((size_t)1 << l) - 1;
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= per * nwin) return;
    size_t k = t / per, r = t % per + 1;
    u32* row = T + (k << w) * ROW;
    Pt<S> A, B, R;
    pt_load<S>(A, row + r * ROW);
    pt_load<S>(B, row + ((size_t)1 << l) * ROW);
    pt_add<S>(R, A, B, E);
    pt_store<S>(row + (((size_t)1 << l) + r) * ROW, R);
}

// K2: out[i] = sum_k T[k][digit_k(e[i])]
template <int S>
__global__ void __launch_bounds__(BLOCK, ECfg<S>::MINW_RUN) k_ec_fixed_exp(u32* __restrict__ out, const u32* __restrict__ T, int w, int nwin,
                                                        const u32* __restrict__ e, int ewords, size_t n, ECDev E) {
    constexpr int ROW = ECfg<S>::ROW;
    size_t el = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (el >= n) return;
    const u32* ep = e + el * ewords;
    Pt<S> A, B;
    u32 d = exp_digit(ep, ewords, 0, w);
    pt_load<S>(A, T + (size_t)d * ROW);
    if (nwin > 1) {                                    // the table is normalised when it is built: the sum runs in XYZZ registers
        PtX<S> R;
        ptx_from_normalised<S>(R, A, E);
#pragma unroll 1
        for (int k = 1; k < nwin; ++k) {
            d = exp_digit(ep, ewords, k * w, w);
            pt_load_normalised<S>(B, T + (((size_t)k << w) + d) * ROW);
            ptx_madd<S>(R, B, E);
        }
        ptx_to_jacobian<S>(A, R, E);
    }
    pt_store<S>(out + el * ROW, A);
}

// Normalisation of an array (Z := 1) so that additions INTO running sums can be mixed (pt_madd).  All Z's are inverted
// together by Montgomery's trick applied level by level: a lane multiplies a chunk of K values up (keeping the running
// products), the chunk products are inverted by the same procedure one level higher, and on the way down a lane turns
// the inverse of its chunk's product into the inverses of its K values.  Only the few values of the top level cost a
// Fermat power (~380 products each); a point costs 1 + 2 products for its inverse, 4 to apply it (zi^2, zi^3, X zi^2,
// Y zi^3) and 3 / K + 3 / K^2 ... for the upper levels.  Points at infinity take part with Z = 1 and stay flagged.
// Field arrays (pref, tot, inv) are FW words per value.
template <int S>
__device__ __forceinline__ void f_store(u32* __restrict__ p, const u32 (&a)[S]) {
    // whole 16-byte words, the padding behind the limbs written as zero (a value owns FW words): three stores instead of ten,
    // and no byte of a cache line left for the memory system to merge
    constexpr int FW = ECfg<S>::FW;
    uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int k = 0; k < FW / 4; ++k) {
        uint4 v;
        v.x = 4 * k + 0 < S ? a[4 * k + 0] : 0;
        v.y = 4 * k + 1 < S ? a[4 * k + 1] : 0;
        v.z = 4 * k + 2 < S ? a[4 * k + 2] : 0;
        v.w = 4 * k + 3 < S ? a[4 * k + 3] : 0;
        q[k] = v;
    }
}
template <int S>
__device__ __forceinline__ void z_of_row(u32 (&z)[S], const u32* __restrict__ row, const ECDev& E) {
    constexpr int FW = ECfg<S>::FW;
    if (row[ECfg<S>::ROW - 1]) {
#pragma unroll
        for (int j = 0; j < S; ++j) z[j] = E.one[j];
    } else {
        f_load<S>(z, row + 2 * FW);
    }
}
// up: pref[i] = the product of the values of i's chunk before i, tot[c] = the product of chunk c.  ROWS: v[i] is the Z of point row i.
// The k arrays of one call go through the two ROW-level kernels in ONE launch (block b works for array b / blocks_per_array,
// as in k_ec_bucket_level): a lane's chunk is a chain of K dependent load + product steps, so a launch lasts one chain whatever
// its size -- seven launches of a third of the device each were seven chains in a row (round 4: 4.1 -> 1.6 ms per pass of configs[4]).
template <int S, bool ROWS>
__global__ void __launch_bounds__(BLOCK, ECfg<S>::MINW) k_finv_up(u32* __restrict__ pref, u32* __restrict__ tot, LevelInputs vs,
                                                                  unsigned blocks_per_array, size_t n, size_t K, ECDev E) {
    constexpr int ROW = ECfg<S>::ROW, FW = ECfg<S>::FW;
    // chunk c = the values c, c + nl, c + 2 nl ... (nl lanes): neighbouring lanes touch neighbouring rows
    const size_t nl = (n + K - 1) / K;
    const unsigned arr = blockIdx.x / blocks_per_array;          // (wave-uniform)
    const u32* __restrict__ v = vs.p[arr];
    pref += (size_t)arr * n * FW;
    tot += (size_t)arr * nl * FW;
    size_t c = (size_t)(blockIdx.x % blocks_per_array) * BLOCK + threadIdx.x;
    if (c >= nl) return;
    // (the loads do not depend on the chain of products: the value of step k + 1 is requested before the product of step k --
    // a launch holds only n / K lanes, two waves per SIMD at 10^6 points, and nothing else would hide the latency)
    u32 acc[S], z[S], zn[S];
#pragma unroll
    for (int j = 0; j < S; ++j) acc[j] = E.one[j];
    auto fetch = [&](u32 (&dst)[S], size_t i) {
        if constexpr (ROWS) z_of_row<S>(dst, v + i * ROW, E);
        else f_load<S>(dst, v + i * FW);
    };
    fetch(z, c);
    for (size_t i = c; i < n; i += nl) {
        const size_t nx = i + nl < n ? i + nl : i;     // (the last step fetches its own value again)
        fetch(zn, nx);
        f_store<S>(pref + i * FW, acc);
        f_mul<S>(acc, acc, z, E);
#pragma unroll
        for (int j = 0; j < S; ++j) z[j] = zn[j];
    }
    f_store<S>(tot + c * FW, acc);
}
// top: inv[i] = 1 / v[i] by Fermat
template <int S>
__global__ void __launch_bounds__(BLOCK, ECfg<S>::MINW) k_finv_top(u32* __restrict__ inv, const u32* __restrict__ v, size_t n, ECDev E) {
    constexpr int FW = ECfg<S>::FW;
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    u32 a[S], r[S];
    f_load<S>(a, v + i * FW);
    f_inv<S>(r, a, E);
    f_store<S>(inv + i * FW, r);
}
// down: inv[i] = 1 / v[i] from invtot[c] = 1 / (the product of chunk c)
template <int S>
__global__ void __launch_bounds__(BLOCK, ECfg<S>::MINW) k_finv_down(u32* __restrict__ inv, const u32* __restrict__ invtot, const u32* __restrict__ pref,
                                                                    const u32* __restrict__ v, size_t n, size_t K, ECDev E) {
    constexpr int FW = ECfg<S>::FW;
    const size_t nl = (n + K - 1) / K;
    size_t c = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (c >= nl) return;
    u32 run[S], pr[S], z[S], zi[S];
    f_load<S>(run, invtot + c * FW);
    const size_t cnt = (n - c + nl - 1) / nl;          // values of this chunk
    for (size_t k = cnt; k-- > 0;) {
        const size_t i = c + k * nl;
        f_load<S>(pr, pref + i * FW);
        f_mul<S>(zi, run, pr, E);
        f_load<S>(z, v + i * FW);
        f_mul<S>(run, run, z, E);
        f_store<S>(inv + i * FW, zi);
    }
}
// the lowest level, fused with the use of the inverses: out[i] = (X zi^2, Y zi^3, 1)
template <int S>
__global__ void __launch_bounds__(BLOCK, ECfg<S>::MINW) k_ec_normalize_down(u32* __restrict__ out, LevelInputs ins, unsigned blocks_per_array,
                                                                            const u32* __restrict__ invtot, const u32* __restrict__ pref,
                                                                            size_t n, size_t K, ECDev E) {
    constexpr int ROW = ECfg<S>::ROW, FW = ECfg<S>::FW;
    const size_t nl = (n + K - 1) / K;
    const unsigned arr = blockIdx.x / blocks_per_array;          // (wave-uniform; see k_finv_up)
    const u32* __restrict__ in = ins.p[arr];
    out += (size_t)arr * n * ROW;
    invtot += (size_t)arr * nl * FW;
    pref += (size_t)arr * n * FW;
    size_t c = (size_t)(blockIdx.x % blocks_per_array) * BLOCK + threadIdx.x;
    if (c >= nl) return;
    u32 run[S];
    f_load<S>(run, invtot + c * FW);
    const size_t cnt = (n - c + nl - 1) / nl;          // points of this chunk
    // one point ahead of the products (see k_finv_up): the prefix and the row of the next step are requested first
    u32 pr[S], prn[S];
    Pt<S> P, Pn;
    if (cnt > 0) {
        const size_t i0 = c + (cnt - 1) * nl;
        f_load<S>(pr, pref + i0 * FW);
        pt_load<S>(P, in + i0 * ROW);
    }
    for (size_t k = cnt; k-- > 0;) {
        const size_t i = c + k * nl;
        const size_t nx = k > 0 ? i - nl : i;          // (the last step fetches its own point again)
        f_load<S>(prn, pref + nx * FW);
        pt_load<S>(Pn, in + nx * ROW);
        u32 zi[S], zi2[S], zi3[S];
        f_mul<S>(zi, run, pr, E);                      // 1 / Z_i
        if (P.inf) {
            pt_set_inf<S>(P, E);                       // (took part with Z = 1: the running inverse is unchanged)
        } else {
            f_mul<S>(run, run, P.Z, E);
            f_sqr<S>(zi2, zi, E);
            f_mul<S>(zi3, zi2, zi, E);
            f_mul<S>(P.X, P.X, zi2, E);
            f_mul<S>(P.Y, P.Y, zi3, E);
#pragma unroll
            for (int j = 0; j < S; ++j) P.Z[j] = E.one[j];
        }
        pt_store<S>(out + i * ROW, P);
        P = Pn;
#pragma unroll
        for (int j = 0; j < S; ++j) pr[j] = prn[j];
    }
}

// K3 product-tree level (see k_bucket_level, also for the arrays of one launch).  FIRST: the inputs are NORMALISED arrays
// (k_ec_normalize), read through `sorted`.
template <int S, bool FIRST>
__global__ void __launch_bounds__(BLOCK, FIRST ? ECfg<S>::MINW_RUN : ECfg<S>::MINW) k_ec_bucket_level(u32* __restrict__ out, size_t out_stride, LevelInputs ins,
                                                           unsigned blocks_per_array,
                                                           const u32* __restrict__ sorted, const u32* __restrict__ off_in,
                                                           const u32* __restrict__ cnt_in, const u32* __restrict__ off_out,
                                                           size_t nbuckets, size_t total_out, u32 F, ECDev E) {
    constexpr int ROW = ECfg<S>::ROW;
    const unsigned arr = blockIdx.x / blocks_per_array;          // (wave-uniform)
    const u32* __restrict__ in = ins.p[arr];
    out += (size_t)arr * out_stride;
    size_t t = (size_t)(blockIdx.x % blocks_per_array) * BLOCK + threadIdx.x;
    if (t >= total_out) return;
    size_t lo = 0, hi = nbuckets;
    while (hi - lo > 1) {
        size_t mid = (lo + hi) >> 1;
        if (off_out[mid] <= t) lo = mid; else hi = mid;
    }
    size_t b = lo;
    u32 j = (u32)(t - off_out[b]);
    u32 start = off_in[b] + j * F;
    u32 end = off_in[b] + cnt_in[b];
    if (end > start + F) end = start + F;
    // FIRST: an entry of `sorted` is the element's index, bit 31 set when its digit in this window is negative (signed
    // windows, vmnhip.hip expprod_words): the point then enters with -Y
    auto row = [&](u32 k) -> const u32* { return FIRST ? in + (size_t)(sorted[k] & 0x7fffffffu) * ROW : in + (size_t)k * ROW; };
    Pt<S> A, B;
    pt_load<S>(A, row(start));
    u32 k0 = start + 1;
    if constexpr (FIRST) {
        if (sorted[start] >> 31) f_neg<S>(A.Y, A.Y, E);
        if (k0 < end) {
            // the chunk's sum runs in XYZZ registers (ptx_madd: 8M + 2S per row); its first addition takes two normalised
            // rows (4M + 2S), the Jacobian row it is stored as costs 2M at the end
            PtX<S> R;
            pt_load_normalised<S>(B, row(k0));
            if (sorted[k0] >> 31) f_neg<S>(B.Y, B.Y, E);
            ptx_mmadd<S>(R, A, B, E);
            // (Running one row ahead of the addition -- the index and the row of k + 1 in flight while row k is added -- was
            // measured and changes nothing, profiles/r04_ec_instruction_diet.txt: the other wave of the SIMD already hides the
            // gather; the kernel is bound by the instructions it issues.)
            for (u32 k = k0 + 1; k < end; ++k) {
                pt_load_normalised<S>(B, row(k));
                if (sorted[k] >> 31) f_neg<S>(B.Y, B.Y, E);
                ptx_madd<S>(R, B, E);
            }
            ptx_to_jacobian<S>(A, R, E);
        }
    } else {
        for (u32 k = k0; k < end; ++k) {
            pt_load<S>(B, row(k));
            pt_add<S>(A, A, B, E);
        }
    }
    pt_store<S>(out + t * ROW, A);
}

// The first level over rows that are NOT normalised (small calls, vmnhip.hip ec_normalise_pays): the gather through `sorted` and
// the signs of k_ec_bucket_level<S, true>, full additions.  Normalising costs a fixed chain of launches with one Fermat power
// at its top (~0.3 ms per call whatever the size); below ~10^5 points the dearer additions are cheaper than that.
template <int S>
__global__ void __launch_bounds__(BLOCK, ECfg<S>::MINW) k_ec_bucket_first_jacobian(u32* __restrict__ out, size_t out_stride, LevelInputs ins,
                                                           unsigned blocks_per_array,
                                                           const u32* __restrict__ sorted, const u32* __restrict__ off_in,
                                                           const u32* __restrict__ cnt_in, const u32* __restrict__ off_out,
                                                           size_t nbuckets, size_t total_out, u32 F, ECDev E) {
    constexpr int ROW = ECfg<S>::ROW;
    const unsigned arr = blockIdx.x / blocks_per_array;          // (wave-uniform)
    const u32* __restrict__ in = ins.p[arr];
    out += (size_t)arr * out_stride;
    size_t t = (size_t)(blockIdx.x % blocks_per_array) * BLOCK + threadIdx.x;
    if (t >= total_out) return;
    size_t lo = 0, hi = nbuckets;
    while (hi - lo > 1) {
        size_t mid = (lo + hi) >> 1;
        if (off_out[mid] <= t) lo = mid; else hi = mid;
    }
    size_t b = lo;
    u32 j = (u32)(t - off_out[b]);
    u32 start = off_in[b] + j * F;
    u32 end = off_in[b] + cnt_in[b];
    if (end > start + F) end = start + F;
    auto load_signed = [&](Pt<S>& P, u32 k) {
        const u32 s = sorted[k];
        pt_load<S>(P, in + (size_t)(s & 0x7fffffffu) * ROW);
        if (s >> 31) {                                 // a negative digit: -Y in the 256p form (Y of a row < 81 p)
            u32 z[S], y[S];
#pragma unroll
            for (int i = 0; i < S; ++i) z[i] = 0;
            f_sub<S, true>(y, z, P.Y, E);
#pragma unroll
            for (int i = 0; i < S; ++i) P.Y[i] = y[i];
        }
    };
    Pt<S> A, B;
    load_signed(A, start);
    for (u32 k = start + 1; k < end; ++k) {
        load_signed(B, k);
        pt_add<S>(A, A, B, E);
    }
    pt_store<S>(out + t * ROW, A);
}

// K5: strided sum (see k_reduce_strided)
template <int S>
__global__ void __launch_bounds__(BLOCK, ECfg<S>::MINW) k_ec_reduce(u32* __restrict__ out, const u32* __restrict__ x, size_t len, size_t Lout,
                                                     size_t nseg, ECDev E) {
    constexpr int ROW = ECfg<S>::ROW;
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= nseg * Lout) return;
    size_t seg = t / Lout, j = t % Lout;
    const u32* base = x + seg * len * ROW;
    Pt<S> A, B;
    pt_load<S>(A, base + j * ROW);
    size_t cnt = (len - j + Lout - 1) / Lout;
    for (size_t k = 1; k < cnt; ++k) {
        pt_load<S>(B, base + (j + k * Lout) * ROW);
        pt_add<S>(A, A, B, E);
    }
    pt_store<S>(out + t * ROW, A);
}

// running sums (the "prods" scan of the modular kernels with + as the operation)
template <int S>
__global__ void __launch_bounds__(BLOCK, ECfg<S>::MINW) k_ec_scan_totals(u32* __restrict__ tot, const u32* __restrict__ e, size_t n, size_t Cc,
                                                          size_t seglen, int rev, ECDev E) {
    constexpr int ROW = ECfg<S>::ROW;
    size_t nchunks = (n + Cc - 1) / Cc;
    size_t c = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (c >= nchunks) return;
    size_t lo = c * Cc, hi = lo + Cc < n ? lo + Cc : n;
    Pt<S> A, B;
    pt_set_inf<S>(A, E);
    for (size_t i = lo; i < hi; ++i) {
        size_t pos = rev ? (i / seglen) * seglen + (seglen - 1 - i % seglen) : i;
        pt_load<S>(B, e + pos * ROW);
        pt_add<S>(A, A, B, E);
    }
    pt_store<S>(tot + c * ROW, A);
}
template <int S>
__global__ void __launch_bounds__(BLOCK, ECfg<S>::MINW) k_ec_scan_apply(u32* __restrict__ out, const u32* __restrict__ e,
                                                         const u32* __restrict__ incoming, size_t n, size_t Cc, size_t seglen,
                                                         int rev, ECDev E) {
    constexpr int ROW = ECfg<S>::ROW;
    size_t nchunks = (n + Cc - 1) / Cc;
    size_t c = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (c >= nchunks) return;
    size_t lo = c * Cc, hi = lo + Cc < n ? lo + Cc : n;
    bool fresh = incoming == nullptr || (lo % seglen) == 0;
    Pt<S> A, B;
    if (fresh) pt_set_inf<S>(A, E);
    else pt_load<S>(A, incoming + (c - 1) * ROW);
    for (size_t i = lo; i < hi; ++i) {
        size_t pos = rev ? (i / seglen) * seglen + (seglen - 1 - i % seglen) : i;
        pt_load<S>(B, e + pos * ROW);
        pt_add<S>(A, A, B, E);
        pt_store<S>(out + pos * ROW, A);
    }
}

// Horner over the window results of a multi-exponentiation: out[a] = sum_w 2^(c w) W[a][w]; one lane per array
// (the chain of c * nwin doublings is sequential, so the k arrays of a multi-array call share its latency)
template <int S>
__global__ void __launch_bounds__(BLOCK) k_ec_horner(u32* __restrict__ out, const u32* __restrict__ wres, int nwin, int c, int k, ECDev E) {
    constexpr int ROW = ECfg<S>::ROW;
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= k) return;
    const u32* wa = wres + (size_t)a * nwin * ROW;
    Pt<S> A, B;
    pt_set_inf<S>(A, E);
    for (int w = nwin - 1; w >= 0; --w) {
        for (int s = 0; s < c; ++s) pt_dbl<S>(A, A, E);
        pt_load<S>(B, wa + (size_t)w * ROW);
        pt_add<S>(A, A, B, E);
    }
    pt_store<S>(out + (size_t)a * ROW, A);
}

}  // namespace vmn
