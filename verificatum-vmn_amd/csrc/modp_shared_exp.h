// modp_shared_exp.h — variants of K1 beside k_modpow: ONE exponent for the whole array (out[i] = x[i]^e, sliding window),
// and two independent exponentiations of small arrays in one launch (k_modpow_jobs, at the end).
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

// k_modpow_shared for arrays of more than one round of tiles: the schedule's steps in phases from a queue of (phase, tile) units
// (see k_modpow_phased, modp_kernels.h: the queue, the hand-over between workgroups and the ONE thread-0 region per turn are the
// same); a tile's table of odd powers lives in a table of its own.  The decryption factors of a party -- one full-length secret
// exponent over every ciphertext -- are this kernel's large case.
template <class C>
__global__ void __launch_bounds__(BLOCK, C::MINW)
k_modpow_shared_phased(u32* __restrict__ out, const u32* __restrict__ x, const SlideStep* __restrict__ steps, int nsteps, int tsize,
                       size_t n, const u32* __restrict__ nmod, u32 n0inv, u32* __restrict__ tab, int phases, u32* __restrict__ queue,
                       u32* __restrict__ done) {
    constexpr int W = C::W;
    extern __shared__ u32 lds[];
    __shared__ u32 s_unit;
    Lane<C> ln(lds);
    u32 nn[C::L];
    load_modulus<C>(nn, nmod, ln);
    const u32 ntiles = (u32)((n + C::EPB - 1) / C::EPB);
    const u32 nunits = ntiles * (u32)phases;
    const int M = nsteps - 1;                            // steps of the main loop (step 0 is the first table read)
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
        u32* mytab = tab + ((size_t)t * C::EPB + ln.eslot) * (size_t)tsize * W;
        u32 a[C::L];
        if (ph == 0) {
            u32 x2[C::L];
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
            load_elem<C>(a, mytab + (size_t)steps[0].idx * W, ln);
        } else {
            load_elem<C>(a, out + ec * W, ln);
        }
        // the steps of this phase: 1 + M ph / P  up to  M (ph + 1) / P
        const int s_lo = 1 + (int)((long)M * ph / phases), s_hi = (int)((long)M * (ph + 1) / phases);
#pragma unroll 1
        for (int s = s_lo; s <= s_hi; ++s) {
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

// Two independent exponentiations in ONE launch (small arrays): job 0 = out0[i] = x0[i]^e0 (one exponent for all), job 1 =
// out1[i] = x1[i]^e1[i] (per-element exponents).  A verifier's check (B) in its separate form needs B^v and B_shift^(k_E): at the
// reference's demo size (10^4 ciphertexts) each of the two kernels alone leaves 40 % of the SIMDs without a wave, and they
// used to run one after the other; as one grid (the first `tiles0` blocks work on job 0) they run side by side.  Fixed window
// `wbits` for both, one tile per block, per-lane tables in `tab` (gridDim.x * EPB * 2^wbits rows).
struct ModpowJob {
    u32* out;
    const u32* x;
    const u32* e;
    int ewords;
    size_t estride;        // words between the exponents of consecutive elements; 0 = one shared exponent
    int ebits;
    size_t n;
};
// one tile (C::EPB elements) of a job; `blocktab`: this block's table rows
template <class C>
__device__ __forceinline__ void modpow_job_tile(const ModpowJob& J, size_t t, int wbits, const u32* __restrict__ nmod, u32 n0inv,
                                                const u32* __restrict__ one_m, u32* __restrict__ blocktab, u32* lds) {
    constexpr int W = C::W;
    Lane<C> ln(lds);
    u32 nn[C::L];
    load_modulus<C>(nn, nmod, ln);
    const int tsize = 1 << wbits;
    u32* mytab = blocktab + (size_t)ln.eslot * (size_t)tsize * W;
    const int nwin = (J.ebits + wbits - 1) / wbits;
    size_t el = t * C::EPB + ln.eslot;
    bool live = el < J.n;
    size_t ec = live ? el : J.n - 1;
    const u32* ep = J.e + ec * J.estride;
    u32 a[C::L];
    load_elem<C>(a, J.x + ec * W, ln);
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
        mont_mul<C>(r, a, ln, nn, n0inv);                            // x * tab[k-1]
        store_elem<C>(mytab + (size_t)k * W, r, ln);
        regs_to_lds<C>(ln, r);
    }
    u32 d = exp_digit(ep, J.ewords, (nwin - 1) * wbits, wbits);
    load_elem<C>(a, mytab + (size_t)d * W, ln);
#pragma unroll 1
    for (int wi = nwin - 2; wi >= 0; --wi) {
#pragma unroll 1
        for (int s = 0; s < wbits; ++s) {
            regs_to_lds<C>(ln, a);
            mont_sqr<C>(a, a, ln, nn, n0inv);
        }
        d = exp_digit(ep, J.ewords, wi * wbits, wbits);
        load_elem_to_lds<C>(ln, mytab + (size_t)d * W);
        mont_mul<C>(a, a, ln, nn, n0inv);
    }
    canonicalize<C>(a, nn, ln);
    if (live) store_elem<C>(J.out + el * W, a, ln);
}
template <class C>
__global__ void __launch_bounds__(BLOCK, C::MINW)
k_modpow_jobs(ModpowJob j0, ModpowJob j1, unsigned tiles0, int wbits, const u32* __restrict__ nmod, u32 n0inv,
              const u32* __restrict__ one_m, u32* __restrict__ tab) {
    extern __shared__ u32 lds[];
    const bool second = blockIdx.x >= tiles0;                        // (block-uniform)
    u32* blocktab = tab + (size_t)blockIdx.x * C::EPB * ((size_t)1 << wbits) * C::W;
    if (second) modpow_job_tile<C>(j1, blockIdx.x - tiles0, wbits, nmod, n0inv, one_m, blocktab, lds);
    else modpow_job_tile<C>(j0, blockIdx.x, wbits, nmod, n0inv, one_m, blocktab, lds);
}
// The same with the two jobs in DIFFERENT geometries of the same rows: the long job eight lanes per element, the short one
// four.  At 10^4 elements each the long chain (766 dependent products at 613 bits) sets the time of the launch; with eight
// lanes a product of that chain takes two thirds of the time, and the short job's four-lane tiles still fit beside it.
template <class CA, class CB>
__global__ void __launch_bounds__(BLOCK, (CA::MINW < CB::MINW ? CA::MINW : CB::MINW))
k_modpow_jobs_mixed(ModpowJob j0, ModpowJob j1, unsigned tiles0, int wbits, const u32* __restrict__ nmod, u32 n0inv,
                    const u32* __restrict__ one_m, u32* __restrict__ tab) {
    static_assert(CA::W == CB::W, "the two geometries read the same rows");
    extern __shared__ u32 lds[];
    constexpr int EPBMAX = CA::EPB > CB::EPB ? CA::EPB : CB::EPB;
    const bool second = blockIdx.x >= tiles0;                        // (block-uniform)
    u32* blocktab = tab + (size_t)blockIdx.x * EPBMAX * ((size_t)1 << wbits) * CA::W;
    if (second) modpow_job_tile<CB>(j1, blockIdx.x - tiles0, wbits, nmod, n0inv, one_m, blocktab, lds);
    else modpow_job_tile<CA>(j0, blockIdx.x, wbits, nmod, n0inv, one_m, blocktab, lds);
}

}  // namespace vmn
