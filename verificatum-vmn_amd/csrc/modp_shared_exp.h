// modp_shared_exp.h — K1b with ONE exponent for the whole array: out[i] = x[i]^e.
//
// The reference raises whole arrays to a single exponent in the decryption half of the mix-net -- the decryption factors
// f = u^(-x_j / c), a full-length secret exponent per party (elgamal/DistrElGamalSession.java:365-385) -- and in the
// verifiers (B^v, raisedu = u^rho: hvzk/PoSBasicTW.java:1028, mixnet/ShufflerElGamalSession.java:884-896).  With the exponent
// shared, every lane takes the same branch at every step, so the left-to-right SLIDING window is free of divergence: the
// host cuts the exponent into odd windows of at most w bits separated by runs of zeros (a list of "square s times, then
// multiply by x^d", d odd), the per-lane table holds the 2^(w-1) odd powers only, and a b-bit exponent costs about
// b / (w + 1) multiplications instead of the b / w of the fixed window k_modpow uses for per-element exponents (2047 bits,
// w = 7: 256 + 64 table products instead of 342 + 62).
#pragma once
#include "modp_kernels.h"

namespace vmn {

// One step of the schedule: `sq` squarings, then a multiplication by the odd power x^(2 idx + 1) (idx < 0: none -- the
// trailing zeros of the exponent).  The first step has sq = 0: the accumulator starts as that power.
struct SlideStep {
    int sq;
    int idx;
};

template <class C>
__global__ void __launch_bounds__(BLOCK, C::MINW)
k_modpow_shared(u32* __restrict__ out, const u32* __restrict__ x, const SlideStep* __restrict__ steps, int nsteps, int tsize, size_t n,
                const u32* __restrict__ nmod, u32 n0inv, u32* __restrict__ tab) {
    constexpr int W = C::W;
    extern __shared__ u32 lds[];
    Lane<C> ln(lds);
    u32 nn[C::L];
    load_modulus<C>(nn, nmod, ln);
    const size_t ntiles = (n + C::EPB - 1) / C::EPB;
    u32* mytab = tab + ((size_t)blockIdx.x * C::EPB + ln.eslot) * (size_t)tsize * W;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        size_t el = t * C::EPB + ln.eslot;
        bool live = el < n;
        size_t ec = live ? el : n - 1;
        u32 a[C::L], x2[C::L];
        // table of odd powers: tab[0] = x, tab[k] = tab[k-1] * x^2
        load_elem<C>(a, x + ec * W, ln);
        store_elem<C>(mytab, a, ln);
        regs_to_lds<C>(ln, a);
        mont_sqr<C>(x2, a, ln, nn, n0inv);
#pragma unroll 1
        for (int k = 1; k < tsize; ++k) {
            regs_to_lds<C>(ln, a);
            u32 r[C::L];
            mont_mul<C>(r, x2, ln, nn, n0inv);         // x^2 * tab[k-1]
            store_elem<C>(mytab + (size_t)k * W, r, ln);
#pragma unroll
            for (int j = 0; j < C::L; ++j) a[j] = r[j];
        }
        // the schedule (wave-uniform: every lane walks the same steps)
        load_elem<C>(a, mytab + (size_t)steps[0].idx * W, ln);
#pragma unroll 1
        for (int s = 1; s < nsteps; ++s) {
            const int sq = steps[s].sq, idx = steps[s].idx;
#pragma unroll 1
            for (int q = 0; q < sq; ++q) {
                regs_to_lds<C>(ln, a);
                mont_sqr<C>(a, a, ln, nn, n0inv);
            }
            if (idx >= 0) {
                load_elem_to_lds<C>(ln, mytab + (size_t)idx * W);
                mont_mul<C>(a, a, ln, nn, n0inv);
            }
        }
        canonicalize<C>(a, nn, ln);
        if (live) store_elem<C>(out + el * W, a, ln);
    }
}

}  // namespace vmn
