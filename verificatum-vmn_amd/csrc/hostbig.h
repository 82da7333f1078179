// hostbig.h — minimal host-side big-integer helpers (little-endian 32-bit words).
//
// Used only for per-modulus / per-base *setup* and for the O(1)-element sequential tails of a
// batched operation (Montgomery constants, the window bases of a fixed-base table, the final
// Horner step of a multi-exponentiation).  Everything that touches all N elements runs on the
// GPU; nothing here is a fallback for a device kernel.  Self-contained on purpose: the product
// library must not link the test oracle (GMP).
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>

namespace vmn {
namespace hostbig {

typedef std::vector<uint32_t> Big;   // little-endian words, fixed length chosen by the caller

inline int cmp(const Big& a, const Big& b) {
    for (size_t i = a.size(); i-- > 0;) {
        if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    }
    return 0;
}
inline bool is_zero(const Big& a) {
    for (uint32_t w : a) if (w) return false;
    return true;
}
// a -= b, returns borrow
inline uint32_t sub_in(Big& a, const Big& b) {
    uint64_t br = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        uint64_t d = (uint64_t)a[i] - b[i] - br;
        a[i] = (uint32_t)d;
        br = (d >> 63) & 1;
    }
    return (uint32_t)br;
}
// a += b, returns carry
inline uint32_t add_in(Big& a, const Big& b) {
    uint64_t c = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        c += (uint64_t)a[i] + b[i];
        a[i] = (uint32_t)c;
        c >>= 32;
    }
    return (uint32_t)c;
}
// a = 2a mod n   (a < n)
inline void dbl_mod(Big& a, const Big& n) {
    uint32_t top = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        uint32_t nt = a[i] >> 31;
        a[i] = (a[i] << 1) | top;
        top = nt;
    }
    if (top || cmp(a, n) >= 0) sub_in(a, n);
}
// big-endian bytes (nbytes) -> words (nw words, zero padded)
inline Big from_be(const uint8_t* be, size_t nbytes, size_t nw) {
    Big r(nw, 0);
    for (size_t i = 0; i < nbytes; ++i) {
        size_t k = nbytes - 1 - i;           // byte significance
        if (k / 4 < nw) r[k / 4] |= (uint32_t)be[i] << (8 * (k % 4));
    }
    return r;
}
inline void to_be(const Big& a, uint8_t* be, size_t nbytes) {
    for (size_t i = 0; i < nbytes; ++i) {
        size_t k = nbytes - 1 - i;
        be[i] = k / 4 < a.size() ? (uint8_t)(a[k / 4] >> (8 * (k % 4))) : 0;
    }
}
inline int bit_length(const Big& a) {
    for (size_t i = a.size(); i-- > 0;) {
        if (a[i]) return (int)(32 * i + 32 - __builtin_clz(a[i]));
    }
    return 0;
}
inline int get_bit(const Big& a, int k) { return (a[k / 32] >> (k % 32)) & 1; }
// floor(a / b) by shift-and-subtract (setup-time only: the cofactor (p - 1) / q of a group); *rem = a mod b
inline Big div(const Big& a, const Big& b, Big* rem = nullptr) {
    const size_t nw = a.size() + 1;
    Big q(a.size(), 0), r(nw, 0), bb(b);
    bb.resize(nw, 0);
    for (int k = bit_length(a) - 1; k >= 0; --k) {
        uint32_t carry = (uint32_t)get_bit(a, k);                    // r = 2 r + bit
        for (size_t i = 0; i < nw; ++i) {
            const uint32_t top = r[i] >> 31;
            r[i] = (r[i] << 1) | carry;
            carry = top;
        }
        if (cmp(r, bb) >= 0) {
            sub_in(r, bb);
            q[k / 32] |= 1u << (k % 32);
        }
    }
    if (rem) {
        r.resize(b.size());
        *rem = r;
    }
    return q;
}

// -n^{-1} mod 2^bits for odd n0 (bits <= 32)
inline uint32_t neg_inv_pow2(uint32_t n0, int bits) {
    uint32_t x = 1;
    for (int i = 0; i < 6; ++i) x *= 2 - n0 * x;    // Newton: doubles the valid bits
    uint32_t mask = bits >= 32 ? 0xffffffffu : ((1u << bits) - 1);
    return (0u - x) & mask;
}

// Montgomery context with 32-bit words on the host: R = 2^(32*nw).
struct Mont {
    Big n;
    size_t nw;
    uint32_t n0inv;   // -n^{-1} mod 2^32
    Big one;          // R mod n
    Big rr;           // R^2 mod n
    explicit Mont(const Big& mod) : n(mod), nw(mod.size()) {
        n0inv = neg_inv_pow2(n[0], 32);
        one.assign(nw, 0);
        one[0] = 1;
        // reduce 1 (it is < n for n > 1), then double 32*nw times
        for (size_t i = 0; i < 32 * nw; ++i) dbl_mod(one, n);
        rr = one;
        for (size_t i = 0; i < 32 * nw; ++i) dbl_mod(rr, n);
    }
    // r = a*b/R mod n (CIOS), canonical
    void mul(Big& r, const Big& a, const Big& b) const {
        std::vector<uint32_t> t(nw + 2, 0);
        for (size_t i = 0; i < nw; ++i) {
            uint64_t c = 0;
            for (size_t j = 0; j < nw; ++j) {
                c += (uint64_t)a[j] * b[i] + t[j];
                t[j] = (uint32_t)c;
                c >>= 32;
            }
            c += t[nw];
            t[nw] = (uint32_t)c;
            t[nw + 1] = (uint32_t)(c >> 32);
            uint32_t m = t[0] * n0inv;
            c = (uint64_t)m * n[0] + t[0];
            c >>= 32;
            for (size_t j = 1; j < nw; ++j) {
                c += (uint64_t)m * n[j] + t[j];
                t[j - 1] = (uint32_t)c;
                c >>= 32;
            }
            c += t[nw];
            t[nw - 1] = (uint32_t)c;
            t[nw] = t[nw + 1] + (uint32_t)(c >> 32);
        }
        Big res(t.begin(), t.begin() + nw);
        if (t[nw] || cmp(res, n) >= 0) sub_in(res, n);
        r.swap(res);
    }
    Big to_mont(const Big& a) const { Big r; mul(r, a, rr); return r; }
    Big from_mont(const Big& a) const { Big o(nw, 0); o[0] = 1; Big r; mul(r, a, o); return r; }
    // base^e mod n in the Montgomery domain (binary, left to right); base_m in Montgomery form
    Big pow_m(const Big& base_m, const Big& e) const {
        Big acc = one;
        for (int k = bit_length(e) - 1; k >= 0; --k) {
            mul(acc, acc, acc);
            if (get_bit(e, k)) mul(acc, acc, base_m);
        }
        return acc;
    }
};

}  // namespace hostbig
}  // namespace vmn
