// light_kernels.h — the kernels that are not templates over an execution geometry: row movement (gather, compare), the
// counting sort and task bookkeeping of the multi-exponentiation, small helpers.  Included by vmnhip.hip only (the
// instantiation units csrc/inst_*.hip include modp_kernels.h, which holds templates only, so that nothing is defined twice).
#pragma once
#include "modp_kernels.h"

namespace vmn {

// t = row * per + c, per a small count of 16-byte chunks per row.  A 64-bit division costs ~100 instructions, more than the
// 16-byte copy it addresses: the 32-bit form is used whenever the index fits (it always does up to 2^32 chunks = 64 GB).
__device__ __forceinline__ void split_chunk(size_t t, int per, size_t& row, int& c) {
    if (t <= 0xffffffffull) {
        const u32 tt = (u32)t, pp = (u32)per;
        const u32 r = tt / pp;
        row = r;
        c = (int)(tt - r * pp);
    } else {
        row = t / (size_t)per;
        c = (int)(t % (size_t)per);
    }
}

// ---- clearing and uploading SMALL things as kernels ------------------------------------------------
// hipMemsetAsync / hipMemcpyAsync go through the runtime's blit path; between kernels of the same stream each of them cost
// 55-70 us of idle device in the small-size timelines (profiles/r04_timeline_p256_n10000_after.txt) where a kernel launch
// costs 7.  A flag word, a histogram of a few thousand counters, a seed or a scalar in a pinned slot are moved by these two
// instead (pinned host memory is mapped into the device's address space: the upload kernel reads it in place).
__global__ void __launch_bounds__(BLOCK) k_zero_words(u32* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) p[i] = 0;
}
__global__ void __launch_bounds__(BLOCK) k_copy_bytes(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) dst[i] = src[i];
}

// ---- generic exclusive scan of a u32 array (three tiny kernels; n up to a few million) ----------
constexpr int SCAN_ITEMS = 16;                         // items per thread

// out[0] = max over the n integers (nw packed little-endian words each) of their bit length.  One thread per
// integer; serves the verifier, which must use every bit of an exponent array it was sent (a reply's k_E) and
// still wants the short path when the entries are as short as an honest prover's.
__global__ void __launch_bounds__(BLOCK) k_words_maxbits(const u32* __restrict__ w, size_t n, int nw, u32* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    u32 bits = 0;
    if (i < n) {
        const u32* p = w + i * (size_t)nw;
        for (int k = nw - 1; k >= 0; --k) {
            u32 v = p[k];
            if (v) {
                bits = 32u * (u32)k + (32u - (u32)__builtin_clz(v));
                break;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) bits = max(bits, (u32)__shfl_xor((int)bits, o));
    if ((threadIdx.x & 63) == 0 && bits) atomicMax(out, bits);
}

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
        size_t row;
        int c;
        split_chunk(t, chunks_per_row, row, c);
        u32 src = idx[row];
        out[t] = src == 0xffffffffu ? fill[c] : in[(size_t)src * chunks_per_row + c];
    }
}

// ---------------------------------------------------------------------------------------------
// K2 fixed base.  Table T[k][d] = base^(d * 2^(w*k)), k < nwin, d < 2^w, rows of W words at
// (k*2^w + d)*W.  The host supplies sq[j] = base^(2^j) (the sequential squaring chain); level l
// fills d in (2^l, 2^(l+1)):  T[k][d] = T[k][d - 2^l] * T[k][2^l].
// ---------------------------------------------------------------------------------------------
// one thread per (k, l) plus the d = 0 rows; plain row copies (W words, constants are in row layout)
__global__ void __launch_bounds__(BLOCK) k_fixed_seed(u32* __restrict__ T, const u32* __restrict__ sq, int w, int nwin,
                                                      const u32* __restrict__ one_row, int W) {
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    size_t total = (size_t)nwin * (w + 1);
    if (t >= total) return;
    int k = (int)(t / (w + 1)), l = (int)(t % (w + 1));
    u32* dst;
    const u32* src;
    if (l == w) {                          // d = 0: the Montgomery one
        dst = T + ((size_t)k << w) * W;
        src = one_row;
    } else {
        dst = T + (((size_t)k << w) + ((size_t)1 << l)) * W;
        src = sq + ((size_t)k * w + l) * W;
    }
    for (int j = 0; j < W; ++j) dst[j] = src[j];
}

// ---------------------------------------------------------------------------------------------
// K3 multi-exponentiation (Pippenger): counting sort of (window, digit), then a product tree per bucket.
// ---------------------------------------------------------------------------------------------
// counts[win][d] += 1 for every element; one thread per (element, window).
// (grid = gx blocks per window: the window of a block is blockIdx.x / gx -- one scalar division per block instead of two
// 64-bit divisions per item, which were most of this kernel's instructions)
// A NARROW window -- the top one when c does not divide the exponent length: 1 bit at c = 15 for 256-bit exponents, all 10^6
// elements on two counters -- is counted in LDS first and reaches the global counters once per block and digit
// (NARROW_BITS; with global atomics alone that one window took longer than all the others together).  The signed windows
// of the curves have 2^12 buckets at 10^6 points: every window of theirs goes this way (2 x 16 KB of LDS in the scatter), and
// the sort of a P-256 multi-exponentiation takes 2.3 ms instead of 5.2 ms.
constexpr int NARROW_BITS = 12;
__device__ __forceinline__ int window_bits(int w, int c, int ebits) {
    const int left = ebits - w * c;
    return left < c ? (left < 1 ? 1 : left) : c;
}
// SIGNED windows (curves, where a negation is free): the c-bit digits are recoded to d' in (-2^(c-1), 2^(c-1)] -- a digit
// above 2^(c-1) becomes d - 2^c and carries one into the next window -- so that a window needs 2^(c-1) buckets instead of 2^c
// (index |d'| - 1; zero digits are not inserted at all) and one more bit per window costs the same aggregation.  The carry into
// window w is decided by the digits below it: the nearest one that is not exactly 2^(c-1) settles it.  ebits covers the
// exponents' storage, so the top window takes the last carry without overflowing.
__device__ __forceinline__ int signed_digit(const u32* __restrict__ ep, int ewords, int w, int c) {
    const u32 half = 1u << (c - 1);
    int carry = 0;
    for (int v = w - 1; v >= 0; --v) {
        const u32 dv = exp_digit(ep, ewords, v * c, c);
        if (dv != half) {
            carry = dv > half;
            break;
        }
    }
    int d = (int)exp_digit(ep, ewords, w * c, c) + carry;
    return d > (int)half ? d - (1 << c) : d;
}
// digit of element i in window w as (bucket index inside the window, negative?), or false for "nothing to insert";
// cb = bucket bits of a window (c, or c - 1 in the signed form)
template <bool SIGNED>
__device__ __forceinline__ bool bucket_of(const u32* __restrict__ ep, int ewords, int w, int c, u32& bucket, u32& neg) {
    if constexpr (SIGNED) {
        const int d = signed_digit(ep, ewords, w, c);
        if (d == 0) return false;
        neg = d < 0;
        bucket = (u32)(d < 0 ? -d : d) - 1u;
        return true;
    } else {
        bucket = exp_digit(ep, ewords, w * c, c);
        neg = 0;
        return true;
    }
}
// distinct bucket indices a window can hold, as a bit count (for the LDS-privatised path)
template <bool SIGNED>
__device__ __forceinline__ int bucket_index_bits(int w, int c, int ebits) {
    if constexpr (SIGNED) {
        const int left = ebits - w * c;
        return left <= 0 ? 0 : (left < c - 1 ? left : c - 1);
    } else {
        return window_bits(w, c, ebits);
    }
}
template <bool SIGNED>
__global__ void __launch_bounds__(BLOCK) k_bucket_hist(u32* __restrict__ counts, const u32* __restrict__ e, int ewords,
                                                       size_t n, int c, int nwin, u32 gx, int ebits) {
    __shared__ u32 h[1 << NARROW_BITS];
    const u32 w = blockIdx.x / gx, bx = blockIdx.x % gx;
    if ((int)w >= nwin) return;
    const int cb = SIGNED ? c - 1 : c;
    u32* __restrict__ cw = counts + ((size_t)w << cb);
    const int bw = bucket_index_bits<SIGNED>((int)w, c, ebits);
    if (bw <= NARROW_BITS) {
        const u32 nd = 1u << bw;
        for (u32 k = threadIdx.x; k < nd; k += BLOCK) h[k] = 0;
        __syncthreads();
        for (size_t i = (size_t)bx * BLOCK + threadIdx.x; i < n; i += (size_t)gx * BLOCK) {
            u32 d, neg;
            if (!bucket_of<SIGNED>(e + i * ewords, ewords, (int)w, c, d, neg)) continue;
            if (d < nd) atomicAdd(&h[d], 1u);
            else atomicAdd(&cw[d], 1u);                // (an exponent wider than the caller declared: still its own bucket)
        }
        __syncthreads();
        for (u32 k = threadIdx.x; k < nd; k += BLOCK)
            if (h[k]) atomicAdd(&cw[k], h[k]);
        return;
    }
    for (size_t i = (size_t)bx * BLOCK + threadIdx.x; i < n; i += (size_t)gx * BLOCK) {
        u32 d, neg;
        if (bucket_of<SIGNED>(e + i * ewords, ewords, (int)w, c, d, neg)) atomicAdd(&cw[d], 1u);
    }
}

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

// sorted[cursor[bucket]++] = element index; one thread per (element, window); bucket = w*2^c + digit.  A narrow window (see
// k_bucket_hist) is placed in two passes over the block's items: count in LDS, reserve one range per digit with a single
// global atomic, then hand out the positions inside the ranges with LDS atomics.
// (SIGNED: the entry carries the digit's sign in bit 31 -- the first tree level negates the point, ec_kernels.h)
template <bool SIGNED>
__global__ void __launch_bounds__(BLOCK) k_bucket_scatter(u32* __restrict__ sorted, u32* __restrict__ cursor,
                                                          const u32* __restrict__ e, int ewords, size_t n, int c, int nwin, u32 gx,
                                                          int ebits) {
    __shared__ u32 h[1 << NARROW_BITS];
    __shared__ u32 base[1 << NARROW_BITS];
    const u32 w = blockIdx.x / gx, bx = blockIdx.x % gx;
    if ((int)w >= nwin) return;
    const int cb = SIGNED ? c - 1 : c;
    u32* __restrict__ cw = cursor + ((size_t)w << cb);
    const int bw = bucket_index_bits<SIGNED>((int)w, c, ebits);
    if (bw <= NARROW_BITS) {
        const u32 nd = 1u << bw;
        for (u32 k = threadIdx.x; k < nd; k += BLOCK) h[k] = 0;
        __syncthreads();
        for (size_t i = (size_t)bx * BLOCK + threadIdx.x; i < n; i += (size_t)gx * BLOCK) {
            u32 d, neg;
            if (bucket_of<SIGNED>(e + i * ewords, ewords, (int)w, c, d, neg) && d < nd) atomicAdd(&h[d], 1u);
        }
        __syncthreads();
        for (u32 k = threadIdx.x; k < nd; k += BLOCK) {
            base[k] = h[k] ? atomicAdd(&cw[k], h[k]) : 0u;
            h[k] = 0;
        }
        __syncthreads();
        for (size_t i = (size_t)bx * BLOCK + threadIdx.x; i < n; i += (size_t)gx * BLOCK) {
            u32 d, neg;
            if (!bucket_of<SIGNED>(e + i * ewords, ewords, (int)w, c, d, neg)) continue;
            u32 pos = d < nd ? base[d] + atomicAdd(&h[d], 1u) : atomicAdd(&cw[d], 1u);
            sorted[pos] = (u32)i | (neg << 31);
        }
        return;
    }
    for (size_t i = (size_t)bx * BLOCK + threadIdx.x; i < n; i += (size_t)gx * BLOCK) {
        u32 d, neg;
        if (!bucket_of<SIGNED>(e + i * ewords, ewords, (int)w, c, d, neg)) continue;
        u32 pos = atomicAdd(&cw[d], 1u);
        sorted[pos] = (u32)i | (neg << 31);
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

// B[b] = the bucket's single remaining item, or one if it is empty (row copies, 16-byte chunks)
__global__ void __launch_bounds__(BLOCK) k_bucket_finalize(uint4* __restrict__ B, const uint4* __restrict__ items,
                                                           const u32* __restrict__ off_in, const u32* __restrict__ cnt_in,
                                                           size_t nbuckets, const uint4* __restrict__ one_row, int cpr) {
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    size_t total = nbuckets * cpr;
    for (; t < total; t += (size_t)gridDim.x * BLOCK) {
        size_t b;
        int ch;
        split_chunk(t, cpr, b, ch);
        B[t] = cnt_in[b] ? items[(size_t)off_in[b] * cpr + ch] : one_row[ch];
    }
}

// overwrite element 0 of every segment with `one` (the d = 0 slot of the suffix products)
__global__ void __launch_bounds__(BLOCK) k_set_segment_heads(uint4* __restrict__ a, size_t seglen, size_t nseg,
                                                             const uint4* __restrict__ one_row, int cpr) {
    size_t t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= nseg * cpr) return;
    size_t s;
    int ch;
    split_chunk(t, cpr, s, ch);
    a[s * seglen * cpr + ch] = one_row[ch];
}

}  // namespace vmn
