// hostnum64.h — host-side big numbers (64-bit limbs) for the O(1) scalars of a proof: ring elements mod q and,
// for ModPGroup, single group elements mod p.  The reference keeps exactly these in VCR scalar classes
// (PRingElement / PGroupElement, e.g. P/hvzk/PoSBasicTW.java:856-888, 1016-1065); arrays never come here.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <vector>

namespace vmn {
namespace num64 {

typedef unsigned __int128 u128;
using Num = std::vector<uint64_t>;      // little-endian limbs
using Bytes = std::vector<uint8_t>;

inline Num from_be(const uint8_t* be, size_t nbytes, size_t nl) {
    Num r(nl, 0);
    for (size_t i = 0; i < nbytes; ++i) {
        size_t k = nbytes - 1 - i;
        if (k / 8 < nl) r[k / 8] |= (uint64_t)be[i] << (8 * (k % 8));
    }
    return r;
}
inline void to_be(const Num& a, uint8_t* be, size_t nbytes) {
    for (size_t i = 0; i < nbytes; ++i) {
        size_t k = nbytes - 1 - i;
        be[i] = k / 8 < a.size() ? (uint8_t)(a[k / 8] >> (8 * (k % 8))) : 0;
    }
}
inline Bytes to_bytes(const Num& a, size_t nbytes) {
    Bytes b(nbytes);
    to_be(a, b.data(), nbytes);
    return b;
}
inline int cmp(const Num& a, const Num& b) {
    for (size_t i = a.size(); i-- > 0;) {
        if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    }
    return 0;
}
inline bool is_zero(const Num& a) {
    for (uint64_t w : a) if (w) return false;
    return true;
}
inline uint64_t add_in(Num& a, const Num& b) {
    u128 c = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        c += (u128)a[i] + b[i];
        a[i] = (uint64_t)c;
        c >>= 64;
    }
    return (uint64_t)c;
}
inline uint64_t sub_in(Num& a, const Num& b) {
    uint64_t borrow = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        u128 d = (u128)a[i] - b[i] - borrow;
        a[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    return borrow;
}
inline int bit_length_be(const uint8_t* be, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        if (be[i]) return (int)(8 * (n - 1 - i) + 32 - __builtin_clz((unsigned)be[i]));
    }
    return 0;
}
inline int bit_length(const Num& a) {
    for (size_t i = a.size(); i-- > 0;) {
        if (a[i]) return (int)(64 * i + 64 - __builtin_clzll(a[i]));
    }
    return 0;
}

// Arithmetic modulo an odd n (Montgomery, R = 2^(64 nl)).  Values handed in and out are plain residues.
struct Mod {
    Num n;
    size_t nl = 0;
    uint64_t n0inv = 0;     // -n^{-1} mod 2^64
    Num one_m, rr;          // R mod n, R^2 mod n

    Mod() {}
    explicit Mod(const Num& modulus) : n(modulus), nl(modulus.size()) {
        uint64_t x = 1;                                   // Newton iteration for n^{-1} mod 2^64
        for (int i = 0; i < 6; ++i) x *= 2 - n[0] * x;
        n0inv = (uint64_t)0 - x;
        one_m.assign(nl, 0);
        one_m[0] = 1;
        for (size_t i = 0; i < 64 * nl; ++i) dbl(one_m);
        rr = one_m;
        for (size_t i = 0; i < 64 * nl; ++i) dbl(rr);
    }
    void dbl(Num& a) const {
        uint64_t top = a[nl - 1] >> 63;
        for (size_t i = nl; i-- > 1;) a[i] = (a[i] << 1) | (a[i - 1] >> 63);
        a[0] <<= 1;
        if (top || cmp(a, n) >= 0) sub_in(a, n);
    }
    // r = a b / R mod n (CIOS); a, b < n
    void mmul(Num& r, const Num& a, const Num& b) const {
        std::vector<uint64_t> t(nl + 2, 0);
        for (size_t i = 0; i < nl; ++i) {
            u128 c = 0;
            const uint64_t bi = b[i];
            for (size_t j = 0; j < nl; ++j) {
                c += (u128)a[j] * bi + t[j];
                t[j] = (uint64_t)c;
                c >>= 64;
            }
            c += t[nl];
            t[nl] = (uint64_t)c;
            t[nl + 1] = (uint64_t)(c >> 64);
            const uint64_t m = t[0] * n0inv;
            c = ((u128)m * n[0] + t[0]) >> 64;
            for (size_t j = 1; j < nl; ++j) {
                c += (u128)m * n[j] + t[j];
                t[j - 1] = (uint64_t)c;
                c >>= 64;
            }
            c += t[nl];
            t[nl - 1] = (uint64_t)c;
            t[nl] = t[nl + 1] + (uint64_t)(c >> 64);
        }
        Num res(t.begin(), t.begin() + nl);
        if (t[nl] || cmp(res, n) >= 0) sub_in(res, n);
        r.swap(res);
    }
    // r = a^2 / R mod n: the cross products once (doubled), then the Montgomery reduction of the 2 nl-limb square --
    // 1.5 nl^2 limb products instead of the 2 nl^2 of mmul.  The sequential squaring chains of the host (a full-length
    // power is ~2000 of them; the chain base^(2^j) of a new fixed-base table likewise) are made of this.
    void msqr(Num& r, const Num& a) const {
        std::vector<uint64_t> t(2 * nl + 1, 0);
        for (size_t i = 0; i + 1 < nl; ++i) {
            u128 c = 0;
            const uint64_t ai = a[i];
            for (size_t j = i + 1; j < nl; ++j) {
                c += (u128)ai * a[j] + t[i + j];
                t[i + j] = (uint64_t)c;
                c >>= 64;
            }
            t[i + nl] = (uint64_t)c;
        }
        for (size_t k = 2 * nl; k-- > 1;) t[k] = (t[k] << 1) | (t[k - 1] >> 63);      // x 2 (the top limb of the cross sum has its high bit clear)
        t[0] <<= 1;
        {
            u128 c = 0;
            for (size_t i = 0; i < nl; ++i) {
                c += (u128)a[i] * a[i] + t[2 * i];
                t[2 * i] = (uint64_t)c;
                c >>= 64;
                c += t[2 * i + 1];
                t[2 * i + 1] = (uint64_t)c;
                c >>= 64;
            }
        }
        for (size_t i = 0; i < nl; ++i) {                                              // REDC, one limb at a time
            const uint64_t m = t[i] * n0inv;
            u128 c = 0;
            for (size_t j = 0; j < nl; ++j) {
                c += (u128)m * n[j] + t[i + j];
                t[i + j] = (uint64_t)c;
                c >>= 64;
            }
            for (size_t k = i + nl; c != 0 && k <= 2 * nl; ++k) {
                c += t[k];
                t[k] = (uint64_t)c;
                c >>= 64;
            }
        }
        Num res(t.begin() + nl, t.begin() + 2 * nl);
        if (t[2 * nl] || cmp(res, n) >= 0) sub_in(res, n);
        r.swap(res);
    }
    Num to_m(const Num& a) const { Num r; mmul(r, a, rr); return r; }
    Num from_m(const Num& a) const { Num o(nl, 0); o[0] = 1; Num r; mmul(r, a, o); return r; }
    Num mul(const Num& a, const Num& b) const { Num t; mmul(t, a, b); Num r; mmul(r, t, rr); return r; }
    Num add(const Num& a, const Num& b) const {
        Num r = a;
        if (add_in(r, b) || cmp(r, n) >= 0) sub_in(r, n);
        return r;
    }
    Num neg(const Num& a) const {
        if (is_zero(a)) return a;
        Num r = n;
        sub_in(r, a);
        return r;
    }
    // big-endian integer of any length -> residue (bitwise Horner; inputs are short, this is not a hot path)
    Num reduce(const uint8_t* be, size_t nbytes) const {
        const int bits = bit_length_be(be, nbytes);
        if (bits <= (int)(64 * nl) && bits <= bit_length(n) + 3) {      // at most a few subtractions away
            Num acc = from_be(be, nbytes, nl);
            while (cmp(acc, n) >= 0) sub_in(acc, n);
            return acc;
        }
        Num acc(nl, 0), o(nl, 0);
        o[0] = 1;
        for (size_t i = 0; i < nbytes; ++i) {
            for (int b = 7; b >= 0; --b) {
                dbl(acc);
                if (((be[i] >> b) & 1) && (add_in(acc, o) || cmp(acc, n) >= 0)) sub_in(acc, n);
            }
        }
        return acc;
    }
    // base^e, e a big-endian non-negative integer of any length; fixed 4-bit windows
    Num pow(const Num& base, const uint8_t* e_be, size_t ebytes) const {
        Num tab[16];
        tab[0] = one_m;
        tab[1] = to_m(base);
        for (int i = 2; i < 16; ++i) mmul(tab[i], tab[i - 1], tab[1]);
        Num acc = one_m;
        bool started = false;
        for (size_t i = 0; i < ebytes; ++i) {
            for (int half = 1; half >= 0; --half) {
                unsigned d = (e_be[i] >> (4 * half)) & 15;
                if (started) {
                    for (int k = 0; k < 4; ++k) msqr(acc, acc);
                }
                if (d) {
                    mmul(acc, acc, tab[d]);
                    started = true;
                }
            }
        }
        return from_m(acc);
    }
    // inverse modulo n (n odd, gcd(a, n) = 1): binary extended Euclid -- O(bits) shift / subtract passes over the limbs
    // (~0.15 ms at 2048 bits; the Fermat power it replaces is a full exponentiation, ~3 ms).  0 -> 0, as before.
    Num inv(const Num& a) const {
        if (is_zero(a)) return a;
        Num u = a, v = n, x1(nl, 0), x2(nl, 0);
        x1[0] = 1;
        auto is_one = [](const Num& x) {
            if (x[0] != 1) return false;
            for (size_t i = 1; i < x.size(); ++i) if (x[i]) return false;
            return true;
        };
        auto shr1 = [](Num& x, uint64_t top) {
            for (size_t i = 0; i + 1 < x.size(); ++i) x[i] = (x[i] >> 1) | (x[i + 1] << 63);
            x[x.size() - 1] = (x[x.size() - 1] >> 1) | (top << 63);
        };
        auto halve_mod = [&](Num& x) {                     // x / 2 mod n
            uint64_t carry = 0;
            if (x[0] & 1) carry = add_in(x, n);
            shr1(x, carry);
        };
        auto sub_mod = [&](Num& x, const Num& y) {         // x - y mod n
            if (sub_in(x, y)) add_in(x, n);
        };
        while (!is_one(u) && !is_one(v)) {
            if (is_zero(u) || is_zero(v)) return Num(nl, 0);            // not invertible
            while (!(u[0] & 1)) {
                shr1(u, 0);
                halve_mod(x1);
            }
            while (!(v[0] & 1)) {
                shr1(v, 0);
                halve_mod(x2);
            }
            if (cmp(u, v) >= 0) {
                sub_in(u, v);
                sub_mod(x1, x2);
            } else {
                sub_in(v, u);
                sub_mod(x2, x1);
            }
        }
        return is_one(u) ? x1 : x2;
    }
    // Jacobi symbol (a / n), n odd, 0 <= a < n: 1, -1, or 0 when gcd(a, n) > 1.  Binary algorithm: shifts and
    // subtractions only.  For a safe prime n = 2q + 1 the symbol is 1 exactly on the subgroup of order q.
    int jacobi(const Num& a_in) const {
        Num a = a_in, m = n;
        int t = 0;
        auto shr = [](Num& x, int k) {
            for (size_t i = 0; i + 1 < x.size(); ++i) x[i] = (x[i] >> k) | (x[i + 1] << (64 - k));
            x[x.size() - 1] >>= k;
        };
        while (!is_zero(a)) {
            if (a[0] == 0) {                                // a whole zero limb: 64 factors of two (even count: no flip)
                for (size_t i = 0; i + 1 < a.size(); ++i) a[i] = a[i + 1];
                a[a.size() - 1] = 0;
                continue;
            }
            int k = __builtin_ctzll(a[0]);
            if (k) {
                const uint64_t m8 = m[0] & 7;
                if ((k & 1) && (m8 == 3 || m8 == 5)) t ^= 1;
                shr(a, k);
            }
            if (cmp(a, m) < 0) {                            // reciprocity
                if ((a[0] & 3) == 3 && (m[0] & 3) == 3) t ^= 1;
                a.swap(m);
            }
            sub_in(a, m);                                   // a >= m, both odd: a - m is even, the symbol unchanged
        }
        for (size_t i = 1; i < m.size(); ++i) if (m[i]) return 0;
        if (m[0] != 1) return 0;
        return t ? -1 : 1;
    }
};

}  // namespace num64
}  // namespace vmn
