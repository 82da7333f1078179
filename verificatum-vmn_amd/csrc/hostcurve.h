// hostcurve.h — single curve points on the host: the O(1) group elements of a proof over ECqPGroup (A', B', C' ...,
// the reference keeps them in VCR's ECqPGroupElement).  Arrays of points never come here (csrc/ec_kernels.h).
#pragma once
#include "hostnum64.h"

namespace vmn {
namespace num64 {

// Short Weierstrass, a = -3, Jacobian coordinates over
// the Montgomery form of Mod.  Points cross as x || y (infinity = all 0xff), the encoding of include/vmnhip.h.
// Inputs are points the GPU import has validated (see check_elements) or results of this code.
struct HostCurve {
    const Mod* F = nullptr;
    size_t cb = 0, fl = 0;                         // coordinate bytes, field limbs
    struct Jac {
        Num X, Y, Z;
        bool inf = true;
    };
    Num sub(const Num& a, const Num& b) const { return F->add(a, F->neg(b)); }
    Num mul(const Num& a, const Num& b) const {
        Num r;
        F->mmul(r, a, b);
        return r;
    }
    Num twice(const Num& a) const { return F->add(a, a); }
    bool decode(const Bytes& e, Num& x, Num& y) const {          // false: the point at infinity
        bool all_ff = true;
        for (uint8_t c : e) all_ff = all_ff && c == 0xff;
        if (all_ff) return false;
        x = F->to_m(from_be(e.data(), cb, fl));
        y = F->to_m(from_be(e.data() + cb, cb, fl));
        return true;
    }
    Bytes encode(const Jac& P) const {
        Bytes out(2 * cb, 0xff);
        if (P.inf || is_zero(P.Z)) return out;
        Num zi = F->to_m(F->inv(F->from_m(P.Z)));
        Num zi2 = mul(zi, zi);
        Num x = F->from_m(mul(P.X, zi2)), y = F->from_m(mul(P.Y, mul(zi2, zi)));
        to_be(x, out.data(), cb);
        to_be(y, out.data() + cb, cb);
        return out;
    }
    Jac dbl(const Jac& P) const {                                // dbl-2001-b
        if (P.inf || is_zero(P.Y)) return Jac();
        Num delta = mul(P.Z, P.Z), gamma = mul(P.Y, P.Y), beta = mul(P.X, gamma);
        Num t = mul(sub(P.X, delta), F->add(P.X, delta));
        Num alpha = F->add(twice(t), t);
        Num beta4 = twice(twice(beta));
        Jac R;
        R.inf = false;
        R.X = sub(mul(alpha, alpha), twice(beta4));
        Num yz = F->add(P.Y, P.Z);
        R.Z = sub(sub(mul(yz, yz), gamma), delta);
        Num g2 = mul(gamma, gamma);
        R.Y = sub(mul(alpha, sub(beta4, R.X)), twice(twice(twice(g2))));
        return R;
    }
    Jac add_affine(const Jac& P, const Num& x2, const Num& y2) const {      // madd with all exceptional cases
        if (P.inf) {
            Jac R;
            R.inf = false;
            R.X = x2;
            R.Y = y2;
            R.Z = F->one_m;
            return R;
        }
        Num z1z1 = mul(P.Z, P.Z);
        Num u2 = mul(x2, z1z1), s2 = mul(y2, mul(P.Z, z1z1));
        Num h = sub(u2, P.X), r = sub(s2, P.Y);
        if (is_zero(h)) {
            if (!is_zero(r)) return Jac();           // P = -Q
            Jac Q;
            Q.inf = false;
            Q.X = x2;
            Q.Y = y2;
            Q.Z = F->one_m;
            return dbl(Q);
        }
        Num hh = mul(h, h), hhh = mul(h, hh), v = mul(P.X, hh);
        Jac R;
        R.inf = false;
        R.X = sub(sub(mul(r, r), hhh), twice(v));
        R.Y = sub(mul(r, sub(v, R.X)), mul(P.Y, hhh));
        R.Z = mul(P.Z, h);
        return R;
    }
    Bytes exp(const Bytes& base, const uint8_t* e_be, size_t ebytes) const {
        Num x, y;
        Jac acc;
        if (!decode(base, x, y)) return encode(acc);
        for (size_t i = 0; i < ebytes; ++i) {
            for (int b = 7; b >= 0; --b) {
                acc = dbl(acc);
                if ((e_be[i] >> b) & 1) acc = add_affine(acc, x, y);
            }
        }
        return encode(acc);
    }
    Bytes add(const Bytes& a, const Bytes& b) const {
        Num x, y;
        Jac acc;
        if (decode(a, x, y)) acc = add_affine(acc, x, y);
        if (decode(b, x, y)) acc = add_affine(acc, x, y);
        return encode(acc);
    }
    Bytes negate(const Bytes& a) const {
        Num x, y;
        if (!decode(a, x, y)) return a;
        Bytes out(a);
        to_be(F->from_m(F->neg(y)), out.data() + cb, cb);
        return out;
    }
};

}  // namespace num64
}  // namespace vmn
