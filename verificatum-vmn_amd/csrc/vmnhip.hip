// vmnhip.hip — the C ABI of include/vmnhip.h over the gfx950 kernels of modp_kernels.h.
//
// Host side only orchestrates: it owns device buffers, picks window sizes, launches kernels on
// the context stream and does the O(1)-element sequential tails (hostbig.h).  There is no CPU
// implementation of any array operation in this library.
#include <stdarg.h>

#include <algorithm>
#include <future>
#include <memory>
#include <system_error>
#include <utility>

#include "ec_kernels.h"
#include "modp_kernels.h"
#include "modp_instances.h"
#include "ec_instances.h"
#include "light_kernels.h"
#include "vmnhip_internal.h"
#include "sha256.h"
#include "sha512.h"
#include "hostnum64.h"

// the Cfg-templated kernels are compiled in the instantiation units csrc/inst_*.hip (modp_instances.h); here they are declared
VMN_UNIT_SMALL(extern template)
VMN_UNIT_2048(extern template)
VMN_UNIT_2048_WIDE(extern template)
VMN_UNIT_3072(extern template)
VMN_UNIT_4096(extern template)
VMN_UNIT_8192(extern template)
VMN_UNIT_16384(extern template)
VMN_UNIT_P224(extern template)
VMN_UNIT_P256(extern template)
VMN_UNIT_P384(extern template)
VMN_UNIT_P521(extern template)

using namespace vmn;
using vmn::hostbig::Big;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void vmn::set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* vmn_last_error(void) { return g_err; }
static thread_local vmn_ctx* tl_helper_of = nullptr;      // the context this thread is the helper of (vmn_ctx_helper_begin)
vmn_ctx* vmn::lane_of(vmn_ctx* c) { return (c && tl_helper_of == c && c->helper) ? c->helper : c; }
extern "C" void vmn_report_error(const char* message) { vmn::set_error("%s", message ? message : ""); }
extern "C" const char* vmn_version(void) { return "vmnhip 0.1 (gfx950, radix-2^28 lazy-carry Montgomery)"; }

#define ARG_CHECK(cond, msg)             \
    do {                                 \
        if (!(cond)) {                   \
            vmn::set_error("%s: %s", __func__, msg); \
            return VMN_ERR_ARG;          \
        }                                \
    } while (0)

// ------------------------------------------------------------------------------------------------
// size dispatch
// ------------------------------------------------------------------------------------------------
// (modulus bits) -> (S limbs, NW words).  One template instantiation per supported size.
// X(limbs, packed words, lanes per element).  3072-bit moduli (110 limbs) run two lanes per element, 4096-bit
// moduli (148 limbs: R = 2^4144) four.
// X(76, 64, 4) and X(112, 96, 4) are the WIDE geometries of 2048- and 3072-bit moduli (modp_kernels.h, struct Cfg): the rows
// of X(74, 64, 1) resp. X(110, 96, 2) worked on by four lanes each; launches over few elements are routed there (geom() below).
// X(80, 64, 8): the widest geometry of 2048-bit moduli (eight lanes per element) for the smallest arrays.
// X(296, 256, 8), X(592, 512, 16): moduli up to 8192 / 16384 bits (the reference offers safe primes up to 15 424 bits,
// demo/mixnet/.conf:189-195, and benchmarks with a 15 492-bit group, benchmarks/bench_config:43): eight / sixteen lanes per
// element, the same kernels; built for completeness, outside north_star's 2048-4096 range and not tuned.
#define VMN_FOR_SIZES(X) X(10, 8, 1) X(14, 12, 1) X(19, 16, 1) X(37, 32, 1) X(74, 64, 1) X(76, 64, 4) X(80, 64, 8) X(110, 96, 2) X(112, 96, 4) X(148, 128, 4) X(296, 256, 8) X(592, 512, 16)
// elliptic curves: X(field limbs, packed words)
// (field limbs are chosen so that R/p >= 2^24: the lazy operand bounds of the point formulas need it)
#define VMN_FOR_CURVES(X) X(9, 7) X(10, 8) X(15, 12) X(21, 17)

static bool size_for_bits(int nbits, int* S, int* NW, int* LPE) {
    const int sizes[][4] = {{256, 10, 8, 1}, {384, 14, 12, 1}, {512, 19, 16, 1}, {1024, 37, 32, 1}, {2048, 74, 64, 1},
                            {3072, 110, 96, 2}, {4096, 148, 128, 4}, {8192, 296, 256, 8}, {16384, 592, 512, 16}};
    for (auto& s : sizes) {
        if (nbits <= s[0]) {
            *S = s[1];
            *NW = s[2];
            *LPE = s[3];
            return true;
        }
    }
    return false;
}

// The geometry a launch over `items` independent elements (chains, tiles ...) runs in: one element per lane fills the chip
// from ~1.3 x 10^5 elements on (2 waves x 4 SIMDs x 256 CUs x 64 lanes); up to ctx->wide_max elements (default 40960: the
// measured crossover of k_modpow / k_fixed_exp, tools/sweep_small_n.sh) the same rows are handed to four lanes each,
// which shortens every chain of dependent products ~2.5 times.  The two geometries share R and the memory layout, so the
// choice is made per launch.  Scans and the tails of reductions are latency-bound at every size (three passes of one
// product per element): they always run wide (`always`).  vmn_ctx_set_small_array_threshold(ctx, 0) turns the wide
// geometry off, a huge value routes everything through it (the test-suite does both); VMN_WIDE_MAX sets the default.
static size_t default_wide_max() {
    static const size_t v = [] {
        const char* env = getenv("VMN_WIDE_MAX");
        return env ? (size_t)strtoull(env, nullptr, 10) : (size_t)40960;
    }();
    return v;
}
// Below ctx->wide8_max elements (default 6144, env VMN_WIDE8_MAX; 2048-bit moduli only) eight lanes share an element: a
// row is 2 x 10 multiply-adds + 22 other instructions instead of 2 x 19 + 16, so a chain is another ~1.2 times shorter --
// as long as every wave still has a SIMD to itself (8 lanes x 8192 elements = 1024 waves); measured crossover ~6000
// elements (tools/sweep_wide8.sh).  The always-wide launches (scans) take this form up to the four-lane threshold.
static size_t default_wide8_max() {
    static const size_t v = [] {
        const char* env = getenv("VMN_WIDE8_MAX");
        return env ? (size_t)strtoull(env, nullptr, 10) : (size_t)6144;
    }();
    return v;
}
// A fixed-base exponentiation over fewer elements than half of this is cut into pieces while n x pieces stays below it
// (env VMN_FIXED_SPLIT_FILL; 0 = never split).  Measured on the PoS leg (profiles/r03_fixed_split_sweep.txt): the family's
// kernel time at N = 10^4 / 4 x 10^4 / 10^5 / 3 x 10^5 is 9.5 / 23.8 / - / - ms unsplit and 4.2 / 13.9 / 41.3 / 108.9 ms with
// 786 432 (47.4 / 110.7 with 196 608): more waves than the two per SIMD that fill the chip still pay, because the kernel waits
// on its random 296-byte table rows.
static size_t fixed_split_fill() {
    static const size_t v = [] {
        const char* env = getenv("VMN_FIXED_SPLIT_FILL");
        return env ? (size_t)strtoull(env, nullptr, 10) : (size_t)786432;
    }();
    return v;
}
static const vmn_modulus& geom(const vmn_ctx* ctx, const vmn_modulus& m, size_t items, bool always = false) {
    const vmn_ctx* root = ctx->parent ? ctx->parent : ctx;
    if (m.wide8 && root->wide8_max != 0 && (items <= root->wide8_max || (always && items <= root->wide_max))) return *m.wide8;
    if (!m.wide || root->wide_max == 0) return m;
    return (always || items <= root->wide_max) ? *m.wide : m;
}
// the non-curve launch sites: X sees `m` = the geometry chosen for `items`
#define VMN_DISPATCH(items, X)                              \
    {                                                       \
        const vmn_modulus& m_base__ = m;                    \
        const vmn_modulus& m = geom(ctx, m_base__, (items)); \
        VMN_FOR_SIZES(X)                                    \
    }

static size_t lds_bytes(const vmn_modulus& m) { return (size_t)m.S * (BLOCK / m.LPE) * sizeof(u32); }
static int blocks_per_cu(const vmn_modulus&) { return 2; }

// ------------------------------------------------------------------------------------------------
// launch helper: dynamic LDS attribute, stream, optional event timing per kernel family
// ------------------------------------------------------------------------------------------------
template <typename... KArgs, typename... Args>
static int launch(vmn_ctx* ctx, const char* family, void (*kernel)(KArgs...), unsigned grid, size_t lds, Args... args) {
    const void* kp = reinterpret_cast<const void*>(kernel);
    if (!ctx->lds_attr_set.count(kp)) {
        VMN_HIP(hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ctx->lds_attr_set.insert(kp);
    }
    TimingRec rec;
    if (ctx->timing) {
        rec.family = family;
        rec.mads = ctx->next_mads;
        rec.canon = ctx->next_canon;
        VMN_HIP(hipEventCreate(&rec.start));
        VMN_HIP(hipEventCreate(&rec.stop));
        VMN_HIP(hipEventRecord(rec.start, ctx->stream));
    }
    ctx->next_mads = ctx->next_canon = 0;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(BLOCK), lds, ctx->stream, static_cast<KArgs>(args)...);
    VMN_HIP(hipGetLastError());
    if (ctx->timing) {
        VMN_HIP(hipEventRecord(rec.stop, ctx->stream));
        ctx->recs.push_back(rec);
    }
    return VMN_OK;
}

// Work accounting for the roofline of the proof legs (bench.py): the number of v_mad_u64_u32 multiply-adds the NEXT
// launch executes, from the number of Montgomery products / squarings its lanes perform.  A product of S limbs is 2 S^2
// multiply-adds (multiplication + reduction half); a squaring S^2 (reduction) + S (S + LPE SQR_BLK) / 2 (block-symmetric
// multiplication half, summed over the LPE lanes of the element).  Curve points: field products
// (S = 10 / 15), 16 per addition (11M + 5S), 8 per doubling (3M + 5S).  Only recorded while timing is on.
static int note_work(vmn_ctx* ctx, const vmn_modulus& m, double products, double squarings = 0, double canon_products = -1) {
    if (!ctx->timing) return 0;
    const double S = m.ec ? (double)m.ec->S : (double)m.S;              // columns
    const double Rw = m.ec ? S : (double)m.rows;                         // rows (< S in a wide geometry)
    // the same products priced in SURVEY.md §8d's unit: 32 x 32-bit multiply-accumulates of a product / squaring on
    // s = ceil(bits / 32) limbs, M(s) = 2 s^2 + s, Q(s) = s (s + 1) / 2 + s^2 + s -- the unit of the headline's roofline
    const double s32 = (double)((m.nbits + 31) / 32);
    // (canon_products: a launch whose formulas execute fewer products than the ones the canonical count was fixed on -- the
    // running sums of the curve kernels, EC_MADD_RUN -- keeps its canonical price)
    ctx->next_canon = (canon_products >= 0 ? canon_products : products) * (2 * s32 * s32 + s32) +
                      squarings * (m.ec ? 2 * s32 * s32 + s32 : s32 * (s32 + 1) / 2 + s32 * s32 + s32);
    if (m.ec) {
        // a field product: S^2 for the multiplication half + S x (non-zero limbs of p, less limb 0 whose carry is folded into
        // column 1) for the reduction rows of the compile-time primes (ec_kernels.h mont_row: 6 of 10 for P-256, 12 of 15 for P-384)
        const double nz = m.ec->S == 10 ? 7 : m.ec->S == 15 ? 12 : S;       // (P-256: 6 limbs + the carry product of the wide-digit rows)
        ctx->next_mads = (products + squarings) * (S * S + S * nz);
        return 0;
    }
    const double sq = Rw * S + Rw * (S + SQR_BLK * m.LPE) / 2;           // block-symmetric in every geometry
    ctx->next_mads = products * 2 * Rw * S + squarings * sq;
    return 0;                                   // (an int so that a launch inside a macro can be written  note_work(..) ? 0 : launch(..))
}
// in field products (S^2 + S nz multiply-adds, see note_work): a product 1, a squaring ~0.775 (symmetric), a zero test 0.5 (reduction only)
// add = 11M + 5S + zero test, mixed add = 7M + 4S + zero test, doubling = 3M + 5S, normalising one point ~7 + inversion / K
static const double EC_ADD = 15.4, EC_MADD = 10.6, EC_DBL = 6.9, EC_NORM = 7.0, EC_INV = 380;
// round 4: a row added into a running sum in XYZZ registers executes 8M + 2S + 2M / run; its canonical price stays EC_MADD
static const double EC_MADD_RUN = 9.7;

static unsigned light_grid(vmn_ctx* ctx, size_t work_items) {
    size_t blocks = (work_items + BLOCK - 1) / BLOCK;
    size_t cap = (size_t)ctx->num_cus * 8;
    return (unsigned)std::max<size_t>(1, std::min(blocks, cap));
}

// plain (non-LDS) kernel launch with timing
template <typename... KArgs, typename... Args>
static int launch_light(vmn_ctx* ctx, const char* family, void (*kernel)(KArgs...), unsigned grid, Args... args) {
    TimingRec rec;
    if (ctx->timing) {
        rec.family = family;
        rec.mads = ctx->next_mads;
        rec.canon = ctx->next_canon;
        VMN_HIP(hipEventCreate(&rec.start));
        VMN_HIP(hipEventCreate(&rec.stop));
        VMN_HIP(hipEventRecord(rec.start, ctx->stream));
    }
    ctx->next_mads = ctx->next_canon = 0;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(BLOCK), 0, ctx->stream, static_cast<KArgs>(args)...);
    VMN_HIP(hipGetLastError());
    if (ctx->timing) {
        VMN_HIP(hipEventRecord(rec.stop, ctx->stream));
        ctx->recs.push_back(rec);
    }
    return VMN_OK;
}

// Device -> host on the lane's stream, complete on return.  Up to STAGE_BYTES through the lane's pinned buffer (a copy into
// pageable memory takes the runtime's slow path: 27 us against 16 us for a verdict word or one element behind a short
// kernel); callers hold the lane's mutex.
constexpr size_t STAGE_BYTES = (size_t)1 << 18;
static int stage_ready(vmn_ctx* ctx) {
    if (ctx->stage) return VMN_OK;
    VMN_HIP(hipHostMalloc(&ctx->stage, STAGE_BYTES, hipHostMallocDefault));
    VMN_HIP(hipEventCreateWithFlags(&ctx->stage_read, hipEventDisableTiming));
    return VMN_OK;
}
static int d2h(vmn_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!bytes) return VMN_OK;
    if (bytes <= STAGE_BYTES) {
        VMN_TRY(stage_ready(ctx));
        // (a host-to-device copy out of the buffer that is still queued is in front of this one on the stream)
        VMN_HIP(hipMemcpyAsync(ctx->stage, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        VMN_HIP(hipStreamSynchronize(ctx->stream));
        ctx->stage_read_pending = false;
        memcpy(dst, ctx->stage, bytes);
        return VMN_OK;
    }
    VMN_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    VMN_HIP(hipStreamSynchronize(ctx->stream));
    return VMN_OK;
}
static int read_flag(vmn_ctx* ctx, uint32_t* out) { return d2h(ctx, out, ctx->flags, sizeof(uint32_t)); }


static int ensure_scratch(vmn_ctx* ctx, size_t bytes) {
    if (ctx->scratch_bytes >= bytes) return VMN_OK;
    if (ctx->scratch) {
        VMN_HIP(hipStreamSynchronize(ctx->stream));
        VMN_HIP(hipFree(ctx->scratch));
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
    }
    VMN_TRACE("scratch:hipMalloc");
    VMN_HIP(hipMalloc(&ctx->scratch, bytes));
    ctx->scratch_bytes = bytes;
    return VMN_OK;
}

// ---- stream-ordered caching allocator ------------------------------------------------------------
// keep at most 96 GB of freed blocks cached (env VMN_POOL_LIMIT_BYTES: several processes sharing one GPU -- a rehearsal of a
// multi-GPU run on one device -- must shrink it, each of them caches on its own)
static const size_t POOL_LIMIT = [] {
    const char* env = getenv("VMN_POOL_LIMIT_BYTES");
    return env && *env ? (size_t)strtoull(env, nullptr, 10) : (size_t)96 << 30;
}();

static size_t block_class_of(vmn_ctx* ctx, const void* p, size_t fallback, bool forget);
// ---- the table arena of a context (vmnhip_internal.h): dropped table blocks are kept for the next table ----------------
static size_t table_arena_limit() {
    static const size_t v = [] {
        const char* env = getenv("VMN_TABLE_ARENA_BYTES");
        return env && *env ? (size_t)strtoull(env, nullptr, 10) : (size_t)48 << 30;
    }();
    return v;
}
// the caller has made sure that no queued work still reads the block
static void table_arena_put(vmn_ctx* lane, void* p, size_t bytes) {
    if (!p) return;
    vmn_ctx* root = lane->parent ? lane->parent : lane;
    std::lock_guard<std::mutex> guard(root->table_arena_mu);
    if (bytes > table_arena_limit()) {
        (void)hipFree(p);
        return;
    }
    root->table_arena.emplace_back(p, bytes);
    root->table_arena_bytes += bytes;
    while (root->table_arena_bytes > table_arena_limit() && !root->table_arena.empty()) {      // oldest first
        (void)hipFree(root->table_arena.front().first);
        root->table_arena_bytes -= root->table_arena.front().second;
        root->table_arena.erase(root->table_arena.begin());
    }
}
// the smallest kept block that holds `bytes` without wasting more than half of itself; nullptr when there is none
static void* table_arena_take(vmn_ctx* lane, size_t bytes) {
    vmn_ctx* root = lane->parent ? lane->parent : lane;
    std::lock_guard<std::mutex> guard(root->table_arena_mu);
    size_t best = root->table_arena.size();
    for (size_t i = 0; i < root->table_arena.size(); ++i) {
        const size_t have = root->table_arena[i].second;
        if (have >= bytes && have <= 2 * bytes && (best == root->table_arena.size() || have < root->table_arena[best].second)) best = i;
    }
    if (best == root->table_arena.size()) return nullptr;
    void* p = root->table_arena[best].first;
    root->table_arena_bytes -= root->table_arena[best].second;
    root->table_arena.erase(root->table_arena.begin() + (long)best);
    return p;
}
static void table_arena_release(vmn_ctx* lane) {
    vmn_ctx* root = lane->parent ? lane->parent : lane;
    std::lock_guard<std::mutex> guard(root->table_arena_mu);
    for (auto& b : root->table_arena) (void)hipFree(b.first);
    root->table_arena.clear();
    root->table_arena_bytes = 0;
}

static void pool_release_all(vmn_ctx* ctx) {
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->pool) {
        for (void* q : kv.second) {
            (void)hipFree(q);
            (void)block_class_of(ctx, q, 0, true);
        }
    }
    ctx->pool.clear();
    ctx->pool_bytes = 0;
}
// Size classes of the pool: multiples of 256 B below 4 KB; above, eight classes per power of two (<= 12.5 % slack).
// Temporaries whose size depends on the data (the level buffers of a multi-exponentiation) would otherwise add a
// new exact size -- and a block that is never reused -- with every call (found by tools/soak.py).
static size_t pool_round(size_t bytes) {
    if (bytes < 256) return 256;
    if (bytes < ((size_t)1 << 12)) return (bytes + 255) & ~(size_t)255;
    int top = 63 - __builtin_clzll((unsigned long long)bytes);
    size_t gran = (size_t)1 << (top - 3);
    return (bytes + gran - 1) & ~(gran - 1);
}
// A request is served by the smallest cached block of its class or of a class up to twice as large: a data-dependent
// size that crosses a class boundary from one proof to the next must not cost a hipMalloc (which maps memory: 20 ms
// were measured for a few MB, as long as all the kernels of a proof of 10^4 ciphertexts).  The block remembers its own
// class (ctx->block_class), so that it returns to the list it came from.
static int pool_alloc(vmn_ctx* ctx, size_t bytes, void** out) {
    bytes = pool_round(bytes);
    for (auto it = ctx->pool.lower_bound(bytes); it != ctx->pool.end() && it->first <= 2 * bytes; ++it) {
        if (it->second.empty()) continue;
        *out = it->second.back();
        it->second.pop_back();
        ctx->pool_bytes -= it->first;
        ctx->live_bytes += it->first;
        return VMN_OK;
    }
    VMN_TRACE("pool:hipMalloc");
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        pool_release_all(ctx);
        e = hipMalloc(out, bytes);
    }
    if (e != hipSuccess) {
        set_error("device allocation of %zu bytes failed: %s", bytes, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? VMN_ERR_NOMEM : VMN_ERR_DEVICE;
    }
    {
        vmn_ctx* root = ctx->parent ? ctx->parent : ctx;         // one table for the lanes of a context: a block may be freed on the other lane
        std::lock_guard<std::mutex> g(root->block_mu);
        root->block_class[*out] = bytes;
    }
    ctx->live_bytes += bytes;
    return VMN_OK;
}
static size_t block_class_of(vmn_ctx* ctx, const void* p, size_t fallback, bool forget) {
    vmn_ctx* root = ctx->parent ? ctx->parent : ctx;
    std::lock_guard<std::mutex> g(root->block_mu);
    auto bc = root->block_class.find(p);
    if (bc == root->block_class.end()) return fallback;
    size_t c = bc->second;
    if (forget) root->block_class.erase(bc);
    return c;
}
static void pool_free(vmn_ctx* ctx, void* p, size_t bytes) {
    if (!p) return;
    const bool drop = ctx->pool_bytes + pool_round(bytes) > POOL_LIMIT;
    bytes = block_class_of(ctx, p, pool_round(bytes), drop);
    ctx->live_bytes -= bytes;
    if (drop) {
        VMN_TRACE("pool:hipFree");
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(p);
        return;
    }
    ctx->pool[bytes].push_back(p);
    ctx->pool_bytes += bytes;
}

// Host<->device copies of small host objects, ordered on the context stream (pool blocks are recycled in
// stream order, so a copy on the null stream could race with kernels still queued on the stream).
// Small ones (exponents, scalars, the index arrays of small permutations) go through the lane's pinned buffer and are only
// QUEUED: the caller's memory is free on return, and the host does not wait for the stream -- a proof at the reference's demo
// size uploads something in front of every other kernel, and each of those waits used to drain the device.
static int h2d(vmn_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!bytes) return VMN_OK;
    if (bytes <= vmn_ctx::UP_SLOT) {                          // the ring of small pinned slots (vmnhip_internal.h)
        if (!ctx->up_ring) {
            VMN_HIP(hipHostMalloc(&ctx->up_ring, vmn_ctx::UP_SLOT * vmn_ctx::UP_SLOTS, hipHostMallocDefault));
            for (unsigned i = 0; i < vmn_ctx::UP_SLOTS; ++i) VMN_HIP(hipEventCreateWithFlags(&ctx->up_done[i], hipEventDisableTiming));
        }
        const unsigned slot = ctx->up_next++ % vmn_ctx::UP_SLOTS;
        if (ctx->up_pending[slot]) VMN_HIP(hipEventSynchronize(ctx->up_done[slot]));      // (32 uploads ago: long done)
        void* at = (char*)ctx->up_ring + (size_t)slot * vmn_ctx::UP_SLOT;
        memcpy(at, src, bytes);
        // (a kernel that reads the pinned slot in place, not hipMemcpyAsync: light_kernels.h k_copy_bytes)
        hipLaunchKernelGGL(k_copy_bytes, dim3((unsigned)((bytes + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, ctx->stream, (uint8_t*)dst,
                           (const uint8_t*)at, bytes);
        VMN_HIP(hipGetLastError());
        VMN_HIP(hipEventRecord(ctx->up_done[slot], ctx->stream));
        ctx->up_pending[slot] = true;
        return VMN_OK;
    }
    if (bytes <= STAGE_BYTES) {
        VMN_TRY(stage_ready(ctx));
        if (ctx->stage_read_pending) VMN_HIP(hipEventSynchronize(ctx->stage_read));     // the previous upload has left the buffer
        memcpy(ctx->stage, src, bytes);
        VMN_HIP(hipMemcpyAsync(dst, ctx->stage, bytes, hipMemcpyHostToDevice, ctx->stream));
        VMN_HIP(hipEventRecord(ctx->stage_read, ctx->stream));
        ctx->stage_read_pending = true;
        return VMN_OK;
    }
    VMN_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    VMN_HIP(hipStreamSynchronize(ctx->stream));          // src is pageable / may go out of scope
    return VMN_OK;
}

// zero `bytes` (a multiple of 4) of device memory on the lane's stream: a kernel up to 4 MB (see k_zero_words), the runtime's
// memset beyond
static int dev_zero(vmn_ctx* ctx, void* p, size_t bytes) {
    if (!bytes) return VMN_OK;
    if (bytes <= ((size_t)4 << 20) && bytes % 4 == 0) {
        const size_t n = bytes / 4;
        const unsigned grid = (unsigned)std::min<size_t>((n + BLOCK - 1) / BLOCK, 1024);
        hipLaunchKernelGGL(k_zero_words, dim3(grid), dim3(BLOCK), 0, ctx->stream, (uint32_t*)p, n);
        VMN_HIP(hipGetLastError());
        return VMN_OK;
    }
    VMN_HIP(hipMemsetAsync(p, 0, bytes, ctx->stream));
    return VMN_OK;
}

// RAII device temporary on the context stream
struct DevTmp {
    vmn_ctx* ctx;
    void* p = nullptr;
    size_t bytes = 0;
    explicit DevTmp(vmn_ctx* c) : ctx(c) {}
    int alloc(size_t nbytes) {          // a second alloc() on the same object hands the first block back (stream-ordered reuse)
        if (p) pool_free(ctx, p, bytes);
        p = nullptr;
        bytes = nbytes ? nbytes : 16;
        return pool_alloc(ctx, bytes, &p);
    }
    ~DevTmp() { pool_free(ctx, p, bytes); }
    DevTmp(const DevTmp&) = delete;
    DevTmp& operator=(const DevTmp&) = delete;
    template <typename T>
    T* as() { return reinterpret_cast<T*>(p); }
};

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
extern "C" int vmn_ctx_create(int device, vmn_ctx** out) {
    ARG_CHECK(out, "null out");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        set_error("vmn_ctx_create: no HIP device available (this library has no CPU fallback)");
        return VMN_ERR_DEVICE;
    }
    ARG_CHECK(device >= 0 && device < count, "device ordinal out of range");
    VMN_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    VMN_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("vmn_ctx_create: device %d is %s, this library is built for gfx950 only", device, prop.gcnArchName);
        return VMN_ERR_DEVICE;
    }
    std::unique_ptr<vmn_ctx> ctx(new vmn_ctx());
    ctx->device = device;
    ctx->num_cus = prop.multiProcessorCount;
    ctx->wide_max = default_wide_max();
    ctx->wide8_max = default_wide8_max();
    VMN_HIP(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;
    VMN_HIP(hipMalloc(&ctx->flags, 64 * sizeof(uint32_t)));
    VMN_HIP(hipMemsetAsync(ctx->flags, 0, 64 * sizeof(uint32_t), ctx->stream));
    *out = ctx.release();
    return VMN_OK;
}

extern "C" void vmn_ctx_destroy(vmn_ctx* ctx) {
    if (!ctx) return;
    if (ctx->helper) {
        vmn_ctx* h = ctx->helper;
        ctx->helper = nullptr;
        if (tl_helper_of == ctx) tl_helper_of = nullptr;
        if (h->order_event) (void)hipEventDestroy(h->order_event);
        vmn_ctx_destroy(h);
    }
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& r : ctx->recs) {
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    pool_release_all(ctx);
    if (!ctx->parent) table_arena_release(ctx);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->flags) (void)hipFree(ctx->flags);
    if (ctx->stage) (void)hipHostFree(ctx->stage);
    if (ctx->stage_read) (void)hipEventDestroy(ctx->stage_read);
    if (ctx->up_ring) {
        (void)hipHostFree(ctx->up_ring);
        for (hipEvent_t e : ctx->up_done)
            if (e) (void)hipEventDestroy(e);
    }
    for (void* sp : ctx->stage_pending)
        if (sp) (void)hipHostFree(sp);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

extern "C" int vmn_ctx_set_stream(vmn_ctx* ctx, void* hip_stream) {
    ARG_CHECK(ctx, "null ctx");
    std::lock_guard<std::recursive_mutex> guard__(ctx->mu);
    VMN_HIP(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return VMN_OK;
}
extern "C" void* vmn_ctx_get_stream(vmn_ctx* ctx) { return ctx ? reinterpret_cast<void*>(ctx->stream) : nullptr; }
extern "C" int vmn_ctx_synchronize(vmn_ctx* ctx) {
    ARG_CHECK(ctx, "null ctx");
    ctx = LANE(ctx);
    std::lock_guard<std::recursive_mutex> guard__(ctx->mu);
    VMN_HIP(hipStreamSynchronize(ctx->stream));
    return VMN_OK;
}

// ---- the helper lane -----------------------------------------------------------------------------------------
static int helper_create(vmn_ctx* ctx) {
    std::lock_guard<std::recursive_mutex> guard__(ctx->mu);
    VMN_HIP(hipSetDevice(ctx->device));
    if (ctx->helper) return VMN_OK;
    std::unique_ptr<vmn_ctx> h(new vmn_ctx());
    h->device = ctx->device;
    h->num_cus = ctx->num_cus;
    h->parent = ctx;
    int least = 0, greatest = 0;                       // numerically lowest = highest priority
    VMN_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    // VMN_HELPER_PRIORITY: "high" (default: a helper's short exports are scheduled between the protocol thread's tiles) or
    // "normal" (the two lanes as equals: what the proof drivers' second lane wants -- measured)
    const char* prio = getenv("VMN_HELPER_PRIORITY");
    const bool normal = prio && (prio[0] == 'n' || prio[0] == '0');
    if (normal) VMN_HIP(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    else VMN_HIP(hipStreamCreateWithPriority(&h->own_stream, hipStreamNonBlocking, greatest));
    h->stream = h->own_stream;
    VMN_HIP(hipMalloc(&h->flags, 64 * sizeof(uint32_t)));
    VMN_HIP(hipMemsetAsync(h->flags, 0, 64 * sizeof(uint32_t), h->stream));
    VMN_HIP(hipEventCreateWithFlags(&h->order_event, hipEventDisableTiming));
    h->timing = ctx->timing;
    ctx->helper = h.release();
    return VMN_OK;
}
// "what the protocol thread has queued so far is what the helper may rely on": an event on the main stream
static int helper_mark_locked(vmn_ctx* ctx) {
    vmn_ctx* h = ctx->helper;
    VMN_HIP(hipEventRecord(h->order_event, ctx->stream));
    h->marked = true;
    return VMN_OK;
}
// order the helper stream behind the latest mark (recording one now if there has never been any)
static int helper_wait_mark(vmn_ctx* ctx) {
    vmn_ctx* h = ctx->helper;
    std::lock_guard<std::mutex> og(h->order_mu);
    if (!h->marked) VMN_TRY(helper_mark_locked(ctx));
    VMN_HIP(hipStreamWaitEvent(h->stream, h->order_event, 0));
    return VMN_OK;
}
extern "C" int vmn_ctx_helper_mark(vmn_ctx* ctx) {
    ARG_CHECK(ctx && !ctx->parent, "null context or a helper lane");
    VMN_TRY(helper_create(ctx));
    VMN_HIP(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> og(ctx->helper->order_mu);
    return helper_mark_locked(ctx);
}
extern "C" int vmn_ctx_helper_begin(vmn_ctx* ctx) {
    ARG_CHECK(ctx && !ctx->parent, "null context or a helper lane");
    VMN_TRY(helper_create(ctx));
    VMN_HIP(hipSetDevice(ctx->device));
    tl_helper_of = ctx;
    return helper_wait_mark(ctx);
}
extern "C" int vmn_ctx_helper_sync(vmn_ctx* ctx) {
    ARG_CHECK(ctx && ctx->helper && tl_helper_of == ctx, "not the helper thread of this context");
    VMN_HIP(hipSetDevice(ctx->device));
    return helper_wait_mark(ctx);
}
extern "C" int vmn_ctx_helper_end(vmn_ctx* ctx) {
    ARG_CHECK(ctx && ctx->helper && tl_helper_of == ctx, "not the helper thread of this context");
    int rc = VMN_OK;
    {
        std::lock_guard<std::recursive_mutex> guard__(ctx->helper->mu);
        if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->helper->stream) != hipSuccess) {
            set_error("vmn_ctx_helper_end: synchronising the helper stream failed");
            rc = VMN_ERR_DEVICE;
        }
    }
    tl_helper_of = nullptr;
    return rc;
}

extern "C" int vmn_ctx_num_cus(vmn_ctx* ctx) { return ctx ? ctx->num_cus : 0; }
extern "C" vmn_ctx* vmn_group_ctx(const vmn_group* grp) { return grp ? grp->ctx : nullptr; }
extern "C" int vmn_ctx_set_small_array_threshold(vmn_ctx* ctx, size_t items) {
    ARG_CHECK(ctx, "null ctx");
    vmn_ctx* root = ctx->parent ? ctx->parent : ctx;
    std::lock_guard<std::recursive_mutex> guard__(root->mu);
    root->wide_max = items;
    return VMN_OK;
}
extern "C" int vmn_ctx_set_tiny_array_threshold(vmn_ctx* ctx, size_t items) {
    ARG_CHECK(ctx, "null ctx");
    vmn_ctx* root = ctx->parent ? ctx->parent : ctx;
    std::lock_guard<std::recursive_mutex> guard__(root->mu);
    root->wide8_max = items;
    return VMN_OK;
}
extern "C" int vmn_ctx_memory_stats(vmn_ctx* ctx, size_t* pool_bytes, size_t* pool_blocks, size_t* live_bytes) {
    ARG_CHECK(ctx, "null ctx");
    size_t blocks = 0, pbytes = 0, live = 0;       // an array may be freed on another lane than it came from: only the sums mean something
    for (vmn_ctx* c : {ctx, ctx->helper}) {
        if (!c) continue;
        std::lock_guard<std::recursive_mutex> guard__(c->mu);
        for (auto& kv : c->pool) blocks += kv.second.size();
        pbytes += c->pool_bytes;
        live += c->live_bytes;
    }
    if (pool_bytes) *pool_bytes = pbytes;
    if (pool_blocks) *pool_blocks = blocks;
    if (live_bytes) *live_bytes = live;
    return VMN_OK;
}

// The launches of BOTH lanes are accounted in the main lane's tables (a driver may run an independent chain of a proof on
// the helper lane: the sum of kernel durations then exceeds the device-busy time, which is what concurrency means).
static int timing_collect_lane(vmn_ctx* into, vmn_ctx* lane) {
    std::lock_guard<std::recursive_mutex> guard__(lane->mu);
    if (lane->recs.empty()) return VMN_OK;
    VMN_HIP(hipStreamSynchronize(lane->stream));
    for (auto& r : lane->recs) {
        float ms = 0;
        VMN_HIP(hipEventElapsedTime(&ms, r.start, r.stop));
        auto& acc = into->timing_acc[r.family];
        acc.first += 1;
        acc.second += ms;
        into->work_acc[r.family] += r.mads;
        into->canon_acc[r.family] += r.canon;
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    lane->recs.clear();
    return VMN_OK;
}
static int timing_collect(vmn_ctx* ctx) {
    VMN_TRY(timing_collect_lane(ctx, ctx));
    if (ctx->helper) VMN_TRY(timing_collect_lane(ctx, ctx->helper));
    return VMN_OK;
}
extern "C" int vmn_ctx_timing_enable(vmn_ctx* ctx, int on) {
    ARG_CHECK(ctx && !ctx->parent, "null ctx or a helper lane");
    std::lock_guard<std::recursive_mutex> guard__(ctx->mu);
    VMN_TRY(timing_collect(ctx));
    ctx->timing = on != 0;
    if (ctx->helper) {
        std::lock_guard<std::recursive_mutex> hguard__(ctx->helper->mu);
        ctx->helper->timing = ctx->timing;
    }
    return VMN_OK;
}
extern "C" int vmn_ctx_timing_reset(vmn_ctx* ctx) {
    ARG_CHECK(ctx, "null ctx");
    std::lock_guard<std::recursive_mutex> guard__(ctx->mu);
    VMN_TRY(timing_collect(ctx));
    ctx->timing_acc.clear();
    ctx->work_acc.clear();
    ctx->canon_acc.clear();
    return VMN_OK;
}
extern "C" int vmn_ctx_timing_get(vmn_ctx* ctx, const char* family, long* launches, double* total_ms) {
    ARG_CHECK(ctx && family, "null argument");
    std::lock_guard<std::recursive_mutex> guard__(ctx->mu);
    VMN_TRY(timing_collect(ctx));
    auto it = ctx->timing_acc.find(family);
    if (launches) *launches = it == ctx->timing_acc.end() ? 0 : it->second.first;
    if (total_ms) *total_ms = it == ctx->timing_acc.end() ? 0.0 : it->second.second;
    return VMN_OK;
}

extern "C" int vmn_ctx_timing_report(vmn_ctx* ctx, char* buf, size_t len) {
    ARG_CHECK(ctx && buf && len > 0, "null argument");
    std::lock_guard<std::recursive_mutex> guard__(ctx->mu);
    VMN_TRY(timing_collect(ctx));
    std::string out;
    for (auto& kv : ctx->timing_acc) {
        char line[160];
        snprintf(line, sizeof(line), "%s %ld %.4f %.6e %.6e\n", kv.first.c_str(), kv.second.first, kv.second.second, ctx->work_acc[kv.first],
                 ctx->canon_acc[kv.first]);
        out += line;
    }
    size_t k = std::min(out.size(), len - 1);
    memcpy(buf, out.data(), k);
    buf[k] = 0;
    return VMN_OK;
}

static void curve_destroy(vmn_curve* c);

// ------------------------------------------------------------------------------------------------
// moduli and groups
// ------------------------------------------------------------------------------------------------
// words (NW, little-endian 32-bit) -> S limbs of 28 bits
// words (NW, little-endian 32-bit) -> one device row: LPE shares of LW words, L limbs each, zero padded
static std::vector<uint32_t> words_to_row_host(const Big& w, int S, int LPE) {
    const int L = S / LPE, LW = stride_for_limbs(L);
    std::vector<uint32_t> row((size_t)LPE * LW, 0);
    for (int j = 0; j < S; ++j) {
        int bit = 28 * j;
        size_t k = bit / 32;
        int sh = bit % 32;
        uint64_t lo = k < w.size() ? w[k] : 0, hi = k + 1 < w.size() ? w[k + 1] : 0;
        row[(size_t)(j / L) * LW + (j % L)] = (uint32_t)(((hi << 32) | lo) >> sh) & LIMB_MASK;
    }
    return row;
}

static int upload_words(vmn_ctx* ctx, uint32_t** dst, const std::vector<uint32_t>& v) {
    VMN_HIP(hipMalloc(dst, v.size() * sizeof(uint32_t)));
    VMN_HIP(hipMemcpy(*dst, v.data(), v.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    (void)ctx;
    return VMN_OK;
}

static void modulus_destroy(vmn_modulus& m) {
    delete m.wide;                         // views: plain data, own nothing
    delete m.wide8;
    if (m.d_n) (void)hipFree(m.d_n);
    if (m.d_rr) (void)hipFree(m.d_rr);
    if (m.d_one) (void)hipFree(m.d_one);
    delete m.hm;
    delete m.hm64;
    m.hm64 = nullptr;
    m = vmn_modulus();
}

// Set up one odd modulus given as big-endian bytes; S/NW are forced (q shares p's geometry).
static int modulus_init(vmn_ctx* ctx, vmn_modulus& m, const uint8_t* be, size_t nbytes, int S, int NW, int LPE) {
    m.S = S;
    m.NW = NW;
    m.LPE = LPE;
    m.rows = S;
    m.W = LPE * stride_for_limbs(S / LPE);
    m.n_words = hostbig::from_be(be, nbytes, NW);
    // bytes beyond NW words must be zero
    for (size_t i = 0; i + (size_t)NW * 4 < nbytes; ++i) {
        if (be[i]) {
            set_error("modulus does not fit %d words", NW);
            return VMN_ERR_ARG;
        }
    }
    m.nbits = hostbig::bit_length(m.n_words);
    if (m.nbits < 16 || !(m.n_words[0] & 1)) {
        set_error("modulus must be odd and at least 16 bits");
        return VMN_ERR_ARG;
    }
    if (m.nbits + 2 > 28 * S) {
        set_error("modulus too large for %d limbs", S);
        return VMN_ERR_ARG;
    }
    m.n0inv = hostbig::neg_inv_pow2(m.n_words[0] & LIMB_MASK, 28);
    // R mod N and R^2 mod N with R = 2^(28 S): double 1 mod N
    Big r(NW, 0);
    r[0] = 1;
    for (int i = 0; i < 28 * S; ++i) hostbig::dbl_mod(r, m.n_words);
    Big rr = r;
    for (int i = 0; i < 28 * S; ++i) hostbig::dbl_mod(rr, m.n_words);
    VMN_TRY(upload_words(ctx, &m.d_n, words_to_row_host(m.n_words, S, LPE)));
    VMN_TRY(upload_words(ctx, &m.d_one, words_to_row_host(r, S, LPE)));
    VMN_TRY(upload_words(ctx, &m.d_rr, words_to_row_host(rr, S, LPE)));
    m.hm = new hostbig::Mont(m.n_words);
    {
        std::vector<uint8_t> nbe((size_t)NW * 4);
        hostbig::to_be(m.n_words, nbe.data(), nbe.size());
        m.hm64 = new num64::Mod(num64::from_be(nbe.data(), nbe.size(), ((size_t)NW * 4 + 7) / 8));
    }
    // the wide geometry of the same rows: Cfg<76, 4> (74 rows, four shares of 19 columns) / Cfg<112, 4> (110 rows, 4 x 28)
    static_assert(Cfg<76, 4>::W == Cfg<74, 1>::W && Cfg<76, 4>::ROWS == 74, "the wide geometry reads the one-lane rows");
    static_assert(Cfg<112, 4>::W == Cfg<110, 2>::W && Cfg<112, 4>::ROWS == 110, "the wide geometry reads the two-lane rows");
    static_assert(Cfg<80, 8>::W == Cfg<74, 1>::W && Cfg<80, 8>::ROWS == 74, "the widest geometry reads the one-lane rows");
    if ((S == 74 && LPE == 1) || (S == 110 && LPE == 2)) {
        m.wide = new vmn_modulus(m);
        m.wide->S = S == 74 ? 76 : 112;
        m.wide->LPE = 4;
        m.wide->wide = nullptr;
        m.wide->wide8 = nullptr;
    }
    if (S == 74 && LPE == 1) {
        m.wide8 = new vmn_modulus(m);
        m.wide8->S = 80;
        m.wide8->LPE = 8;
        m.wide8->wide = nullptr;
        m.wide8->wide8 = nullptr;
    }
    return VMN_OK;
}

extern "C" int vmn_modp_group_create(vmn_ctx* ctx, const uint8_t* p_be, const uint8_t* q_be, const uint8_t* g_be,
                                     size_t nbytes, vmn_group** out) {
    ARG_CHECK(ctx && p_be && q_be && g_be && out && nbytes > 0, "null argument");
    VMN_ENTER(ctx);
    Big pw = hostbig::from_be(p_be, nbytes, (nbytes + 3) / 4);
    int nbits = hostbig::bit_length(pw);
    int S, NW, LPE;
    if (!size_for_bits(nbits, &S, &NW, &LPE)) {
        set_error("vmn_modp_group_create: %d-bit modulus not supported (max 16384)", nbits);
        return VMN_ERR_UNSUPPORTED;
    }
    std::unique_ptr<vmn_group> g(new vmn_group());
    g->ctx = ctx;
    g->nbytes = nbytes;
    g->xbytes = nbytes;
    int rc = modulus_init(ctx, g->P, p_be, nbytes, S, NW, LPE);
    if (rc == VMN_OK) rc = modulus_init(ctx, g->Q, q_be, nbytes, S, NW, LPE);
    if (rc != VMN_OK) {
        modulus_destroy(g->P);
        modulus_destroy(g->Q);
        return rc;
    }
    g->g_words = hostbig::from_be(g_be, nbytes, NW);
    *out = g.release();
    return VMN_OK;
}

extern "C" void vmn_group_destroy(vmn_group* grp) {
    if (!grp) return;
    std::lock_guard<std::recursive_mutex> guard__(grp->ctx->mu);
    (void)hipStreamSynchronize(grp->ctx->stream);
    if (grp->ctx->helper) (void)hipStreamSynchronize(grp->ctx->helper->stream);
    for (auto& kv : grp->fixed) {
        table_arena_put(grp->ctx, kv.second.d_tab, kv.second.bytes);      // (both streams have drained: nothing reads them any more)
        if (kv.second.ready) (void)hipEventDestroy(kv.second.ready);
    }
    if (grp->curve) {
        if (grp->P.d_one) (void)hipFree(grp->P.d_one);
        grp->P = vmn_modulus();
        curve_destroy(grp->curve);
    } else {
        modulus_destroy(grp->P);
    }
    modulus_destroy(grp->Q);
    delete grp;
}
extern "C" size_t vmn_group_elem_bytes(const vmn_group* grp) { return grp ? (grp->curve ? 2 * grp->nbytes : grp->nbytes) : 0; }
extern "C" size_t vmn_group_exp_bytes(const vmn_group* grp) { return grp ? grp->xbytes : 0; }
// Java's BigInteger.toByteArray().length of a positive integer of `bits` bits: one sign bit on top
static size_t java_width(int bits) { return (size_t)bits / 8 + 1; }
extern "C" int vmn_group_set_wire_bytes(vmn_group* grp, size_t elem_bytes, size_t exp_bytes) {
    ARG_CHECK(grp, "null group");
    VMN_ENTER(LANE(grp->ctx));
    std::lock_guard<std::recursive_mutex> tab_guard(grp->tab_mu);
    ARG_CHECK(grp->fixed.empty(), "wire widths must be chosen before the group is used");
    const int pbits = grp->curve ? hostbig::bit_length(grp->curve->p_words) : grp->P.nbits;
    size_t eb = elem_bytes ? elem_bytes : java_width(pbits);
    size_t xb = exp_bytes ? exp_bytes : java_width(grp->Q.nbits);
    ARG_CHECK(8 * eb >= (size_t)pbits && 8 * xb >= (size_t)grp->Q.nbits, "width too small for the modulus / order");
    ARG_CHECK(eb <= 4096 && xb <= 4096, "width too large");
    grp->nbytes = eb;
    grp->xbytes = xb;
    return VMN_OK;
}
extern "C" int vmn_group_kind(const vmn_group* grp) { return grp && grp->curve ? 1 : 0; }
extern "C" size_t vmn_group_table_bytes(const vmn_group* grp) { return grp ? grp->fixed_bytes : 0; }
extern "C" int vmn_group_get_order(const vmn_group* grp, uint8_t* q_be) {
    ARG_CHECK(grp && q_be, "null argument");
    hostbig::to_be(grp->Q.n_words, q_be, grp->xbytes);
    return VMN_OK;
}
extern "C" int vmn_group_get_modulus(const vmn_group* grp, uint8_t* p_be) {
    ARG_CHECK(grp && p_be, "null argument");
    hostbig::to_be(grp->curve ? grp->curve->p_words : grp->P.n_words, p_be, grp->nbytes);
    return VMN_OK;
}
extern "C" int vmn_group_get_generator(const vmn_group* grp, uint8_t* g_be) {
    ARG_CHECK(grp && g_be, "null argument");
    if (grp->curve) {
        hostbig::to_be(grp->curve->gx_words, g_be, grp->nbytes);
        hostbig::to_be(grp->curve->gy_words, g_be + grp->nbytes, grp->nbytes);
    } else {
        hostbig::to_be(grp->g_words, g_be, grp->nbytes);
    }
    return VMN_OK;
}


// ------------------------------------------------------------------------------------------------
// elliptic-curve groups
// ------------------------------------------------------------------------------------------------
static ECDev ecdev(const vmn_curve* c) {
    ECDev E;
    E.p = c->d_p;
    E.one = c->d_one;
    E.rr = c->d_rr;
    E.b = c->d_b;
    E.mp = c->d_mp;
    E.mp2 = c->d_mp2;
    E.pm2 = c->d_pm2;
    E.pp14 = c->d_pp14;
    E.n0inv = c->n0inv;
    E.p1p = c->p1p;
    E.c16 = 16;
    E.pwords = c->NW;
    E.ts_s = c->ts_s;
    E.ts_ewords = c->ts_ewords;
    E.ts_e = c->d_ts_e;
    E.ts_c = c->d_ts_c;
    return E;
}

struct CurveParams {
    const char* name;
    int bits;
    int field_limbs;   // 28-bit limbs of a field element: R / p >= 2^24 (the lazy operand bounds of the point formulas)
    int words;         // 32-bit words of a coordinate
    const char* p;
    const char* n;
    const char* b;
    const char* gx;
    const char* gy;
};
static const CurveParams kCurves[] = {
    {"P-224", 224, 9, 7, "ffffffffffffffffffffffffffffffff000000000000000000000001",
     "ffffffffffffffffffffffffffff16a2e0b8f03e13dd29455c5c2a3d",
     "b4050a850c04b3abf54132565044b0b7d7bfd8ba270b39432355ffb4",
     "b70e0cbd6bb4bf7f321390b94a03c1d356c21122343280d6115c1d21",
     "bd376388b5f723fb4c22dfe6cd4375a05a07476444d5819985007e34"},
    {"P-256", 256, 10, 8, "ffffffff00000001000000000000000000000000ffffffffffffffffffffffff",
     "ffffffff00000000ffffffffffffffffbce6faada7179e84f3b9cac2fc632551",
     "5ac635d8aa3a93e7b3ebbd55769886bc651d06b0cc53b0f63bce3c3e27d2604b",
     "6b17d1f2e12c4247f8bce6e563a440f277037d812deb33a0f4a13945d898c296",
     "4fe342e2fe1a7f9b8ee7eb4a7c0f9e162bce33576b315ececbb6406837bf51f5"},
    {"P-384", 384, 15, 12,
     "fffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffeffffffff0000000000000000ffffffff",
     "ffffffffffffffffffffffffffffffffffffffffffffffffc7634d81f4372ddf581a0db248b0a77aecec196accc52973",
     "b3312fa7e23ee7e4988e056be3f82d19181d9c6efe8141120314088f5013875ac656398d8a2ed19d2a85c8edd3ec2aef",
     "aa87ca22be8b05378eb1c71ef320ad746e1d3b628ba79b9859f741e082542a385502f25dbf55296c3a545e3872760ab7",
     "3617de4a96262c6f5d9e98bf9292dc29f8f41dbd289a147ce9da3113b5f0b8c00a60b1ce1d7e819d7a431d7c90ea0e5f"},
    // (odd hex lengths are padded with a leading zero nibble: 521 bits = 66 bytes on the wire)
    {"P-521", 521, 21, 17,
     "01ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff",
     "01fffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffa51868783bf2f966b7fcc0148f709a5d03bb5c9b8899c47aebb6fb71e91386409",
     "0051953eb9618e1c9a1f929a21a0b68540eea2da725b99b315f3b8b489918ef109e156193951ec7e937b1652c0bd3bb1bf073573df883d2c34f1ef451fd46b503f00",
     "00c6858e06b70404e9cd9e3ecb662395b4429c648139053fb521f828af606b4d3dbaa14b5e77efe75928fe1dc127a2ffa8de3348b3c1856a429bf97e7e31c2e5bd66",
     "011839296a789a3bc0045c8a5fb42c7d1bd998f54449579b446817afbd17273e662c97ee72995ef42640c550b9013fad0761353c7086a272c24088be94769fd16650"},
};

static std::vector<uint8_t> hex_to_be(const char* h) {
    size_t n = strlen(h) / 2;
    std::vector<uint8_t> out(n);
    auto nib = [](char c) -> int { return c <= '9' ? c - '0' : (c | 32) - 'a' + 10; };
    for (size_t i = 0; i < n; ++i) out[i] = (uint8_t)(nib(h[2 * i]) * 16 + nib(h[2 * i + 1]));
    return out;
}
// x * 2^k mod n by repeated doubling (x < n)
static Big shift_mod(Big x, int k, const Big& n) {
    for (int i = 0; i < k; ++i) hostbig::dbl_mod(x, n);
    return x;
}
// limbs (28-bit) of an arbitrary word vector, S limbs, no reduction
static std::vector<uint32_t> limbs_of(const Big& w, int S) {
    std::vector<uint32_t> l(S, 0);
    for (int j = 0; j < S; ++j) {
        int bit = 28 * j;
        size_t k = bit / 32;
        int sh = bit % 32;
        uint64_t lo = k < w.size() ? w[k] : 0, hi = k + 1 < w.size() ? w[k + 1] : 0;
        l[j] = (uint32_t)(((hi << 32) | lo) >> sh) & LIMB_MASK;
    }
    return l;
}
static Big times_small(const Big& a, uint32_t k, size_t nw) {
    Big r(nw, 0);
    uint64_t c = 0;
    for (size_t i = 0; i < nw; ++i) {
        c += (uint64_t)(i < a.size() ? a[i] : 0) * k;
        r[i] = (uint32_t)c;
        c >>= 32;
    }
    return r;
}

static void curve_destroy(vmn_curve* c) {
    if (!c) return;
    if (c->d_consts) (void)hipFree(c->d_consts);
    delete c->f64;
    delete c;
}

static int curve_create(vmn_ctx* ctx, const CurveParams& cp, vmn_curve** out) {
    std::unique_ptr<vmn_curve> c(new vmn_curve());
    c->name = cp.name;
    int S, NW, LPE;
    if (!size_for_bits(cp.bits, &S, &NW, &LPE) || LPE != 1) return VMN_ERR_UNSUPPORTED;
    S = cp.field_limbs;
    NW = cp.words;
    c->S = S;
    c->NW = NW;
    auto pb = hex_to_be(cp.p), bb = hex_to_be(cp.b), gx = hex_to_be(cp.gx), gy = hex_to_be(cp.gy);
    c->p_words = hostbig::from_be(pb.data(), pb.size(), NW);
    c->b_words = hostbig::from_be(bb.data(), bb.size(), NW);
    c->gx_words = hostbig::from_be(gx.data(), gx.size(), NW);
    c->gy_words = hostbig::from_be(gy.data(), gy.size(), NW);
    c->n0inv = hostbig::neg_inv_pow2(c->p_words[0] & LIMB_MASK, 28);
    c->f64 = new num64::Mod(num64::from_be(pb.data(), pb.size(), (pb.size() + 7) / 8));
    c->host.F = c->f64;
    c->host.cb = ((size_t)cp.bits + 7) / 8;
    c->host.fl = c->f64->nl;
    {
        const num64::Mod& F = *c->f64;
        num64::Num two28(F.nl, 0), rd(F.nl, 0);
        two28[0] = (uint64_t)1 << 28;
        rd[0] = 1;
        for (int j = 0; j < c->S; ++j) rd = F.mul(rd, two28);
        c->rd_inv = F.inv(rd);
    }
    const Big& pw = c->p_words;
    c->p1p = limbs_of(pw, S)[1] + 1;
    {   // the kernels carry the primes of these two sizes as compile-time constants (ec_kernels.h FieldPrime): same prime?
        const std::vector<uint32_t> lim = limbs_of(pw, S);
        bool same = c->n0inv == 1;
        if (S == 10) for (int j = 0; j < S; ++j) same = same && lim[j] == FieldPrime<10>::limb[j];
        else if (S == 15) for (int j = 0; j < S; ++j) same = same && lim[j] == FieldPrime<15>::limb[j];
        if ((S == 10 || S == 15) && !same) {
            set_error("curve %s: a field prime of %d limbs other than the one compiled into the kernels", cp.name, S);
            return VMN_ERR_UNSUPPORTED;
        }
    }
    Big one(NW, 0);
    one[0] = 1;
    Big r1 = shift_mod(one, 28 * S, pw);                 // R mod p
    Big r2 = shift_mod(r1, 28 * S, pw);                  // R^2 mod p
    Big bm = shift_mod(c->b_words, 28 * S, pw);          // b R mod p
    Big pm2 = pw;
    Big two(NW, 0);
    two[0] = 2;
    hostbig::sub_in(pm2, two);
    const int FW = stride_for_limbs(S);
    std::vector<uint32_t> blob;
    auto put = [&](const std::vector<uint32_t>& v, int words) {
        size_t off = blob.size();
        blob.insert(blob.end(), v.begin(), v.end());
        blob.resize(off + words, 0);
        return off;
    };
    size_t o_p = put(limbs_of(pw, S), FW), o_one = put(limbs_of(r1, S), FW), o_rr = put(limbs_of(r2, S), FW);
    size_t o_b = put(limbs_of(bm, S), FW);
    size_t o_mp = put(limbs_of(times_small(pw, 64, NW + 1), S), FW);
    size_t o_mp2 = put(limbs_of(times_small(pw, 256, NW + 1), S), FW);
    size_t o_pm2 = put(std::vector<uint32_t>(pm2.begin(), pm2.end()), (NW + 3) & ~3);
    Big pp14(NW + 1, 0);                                  // (p + 1) / 4: the square-root exponent when p = 3 mod 4
    {
        uint64_t c = 1;
        for (int i = 0; i < NW; ++i) {
            c += pw[i];
            pp14[i] = (uint32_t)c;
            c >>= 32;
        }
        pp14[NW] = (uint32_t)c;
        for (int i = 0; i < NW; ++i) pp14[i] = (pp14[i] >> 2) | (pp14[i + 1] << 30);
        pp14.resize(NW);
    }
    size_t o_pp14 = put(std::vector<uint32_t>(pp14.begin(), pp14.end()), (NW + 3) & ~3);
    // Square roots when p = 1 mod 4 (P-224): Tonelli-Shanks constants -- p - 1 = 2^s Q, the exponent (Q - 1) / 2, and
    // c = z^Q for the smallest non-residue z (Euler's criterion on the host), as field limbs in Montgomery form.
    size_t o_tse = 0, o_tsc = 0;
    c->ts_s = 0;
    c->ts_ewords = 0;
    if ((pw[0] & 3u) == 1u) {
        Big q = pw;
        q[0] -= 1;                                                  // p - 1
        int s2 = 0;
        while (!hostbig::get_bit(q, 0)) {
            for (int i = 0; i < NW; ++i) q[i] = (q[i] >> 1) | (i + 1 < NW ? q[i + 1] << 31 : 0);
            ++s2;
        }
        Big half = pw;                                              // (p - 1) / 2
        half[0] -= 1;
        for (int i = 0; i < NW; ++i) half[i] = (half[i] >> 1) | (i + 1 < NW ? half[i + 1] << 31 : 0);
        hostbig::Mont hm(pw);
        Big zc;
        for (uint32_t z = 2; z < 200 && zc.empty(); ++z) {
            Big zb(NW, 0);
            zb[0] = z;
            const Big zm = hm.to_mont(zb);
            if (hostbig::cmp(hm.pow_m(zm, half), hm.one) != 0) zc = hm.from_mont(hm.pow_m(zm, q));     // z^((p-1)/2) = -1: a non-residue
        }
        if (zc.empty()) {
            set_error("curve %s: no quadratic non-residue found", cp.name);
            return VMN_ERR_UNSUPPORTED;
        }
        Big e = q;                                                  // (Q - 1) / 2
        for (int i = 0; i < NW; ++i) e[i] = (e[i] >> 1) | (i + 1 < NW ? e[i + 1] << 31 : 0);
        c->ts_s = s2;
        c->ts_ewords = (hostbig::bit_length(e) + 31) / 32;
        if (c->ts_ewords < 1) c->ts_ewords = 1;
        o_tse = put(std::vector<uint32_t>(e.begin(), e.end()), (NW + 3) & ~3);
        o_tsc = put(limbs_of(shift_mod(zc, 28 * S, pw), S), FW);   // z^Q R mod p
    }
    VMN_TRY(upload_words(ctx, &c->d_consts, blob));
    c->d_ts_e = c->d_consts + o_tse;
    c->d_ts_c = c->d_consts + o_tsc;
    c->d_p = c->d_consts + o_p;
    c->d_one = c->d_consts + o_one;
    c->d_rr = c->d_consts + o_rr;
    c->d_b = c->d_consts + o_b;
    c->d_mp = c->d_consts + o_mp;
    c->d_mp2 = c->d_consts + o_mp2;
    c->d_pm2 = c->d_consts + o_pm2;
    c->d_pp14 = c->d_consts + o_pp14;
    *out = c.release();
    return VMN_OK;
}

extern "C" int vmn_ec_group_create(vmn_ctx* ctx, const char* curve_name, vmn_group** out) {
    ARG_CHECK(ctx && curve_name && out, "null argument");
    VMN_ENTER(ctx);
    const CurveParams* cp = nullptr;
    for (auto& k : kCurves) {
        if (strcmp(k.name, curve_name) == 0) cp = &k;
    }
    if (!cp) {
        set_error("vmn_ec_group_create: unknown curve %s (known: P-224, P-256, P-384, P-521)", curve_name);
        return VMN_ERR_UNSUPPORTED;
    }
    std::unique_ptr<vmn_group> g(new vmn_group());
    g->ctx = ctx;
    g->nbytes = ((size_t)cp->bits + 7) / 8;
    g->xbytes = g->nbytes;
    vmn_curve* curve = nullptr;
    VMN_TRY(curve_create(ctx, *cp, &curve));
    g->curve = curve;
    // scalars: ordinary residues mod the group order
    auto nb = hex_to_be(cp->n);
    int S, NW, LPE;
    size_for_bits(cp->bits, &S, &NW, &LPE);
    int rc = modulus_init(ctx, g->Q, nb.data(), nb.size(), S, NW, LPE);
    if (rc != VMN_OK) {
        curve_destroy(curve);
        return rc;
    }
    // point rows: 3 field elements; the identity row is (one, one, 0 | flag)
    vmn_modulus& P = g->P;
    P.S = curve->S;
    P.NW = curve->NW;
    P.LPE = 1;
    P.W = 3 * stride_for_limbs(curve->S);
    P.nbits = cp->bits;
    P.ec = curve;
    P.n_words = curve->p_words;
    {
        const int FW = stride_for_limbs(curve->S);
        std::vector<uint32_t> one(curve->S), row(P.W, 0);
        hipError_t he = hipMemcpy(one.data(), curve->d_one, curve->S * sizeof(uint32_t), hipMemcpyDeviceToHost);
        if (he != hipSuccess) {
            set_error("curve constants readback failed: %s", hipGetErrorString(he));
            modulus_destroy(g->Q);
            curve_destroy(curve);
            return VMN_ERR_DEVICE;
        }
        for (int j = 0; j < curve->S; ++j) {
            row[j] = one[j];
            row[FW + j] = one[j];
        }
        row[P.W - 1] = 1;
        rc = upload_words(ctx, &P.d_one, row);
        if (rc != VMN_OK) {
            modulus_destroy(g->Q);
            curve_destroy(curve);
            return rc;
        }
    }
    *out = g.release();
    return VMN_OK;
}

// ------------------------------------------------------------------------------------------------
// generic array plumbing (group arrays are residues mod p, ring arrays residues mod q)
// ------------------------------------------------------------------------------------------------
static size_t elem_words(const vmn_modulus& m) { return (size_t)m.W; }
static unsigned grid_for(size_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK); }                      // one thread per item
static unsigned egrid(const vmn_modulus& m, size_t n) {                                                    // one lane (pair) per element
    size_t epb = BLOCK / m.LPE;
    return (unsigned)((n + epb - 1) / epb);
}

static size_t elems_bytes(const vmn_modulus& m, size_t n) { return std::max<size_t>(n, 1) * elem_words(m) * sizeof(uint32_t); }
static int alloc_elems(vmn_ctx* ctx, const vmn_modulus& m, size_t n, uint32_t** d) {
    return pool_alloc(ctx, elems_bytes(m, n), reinterpret_cast<void**>(d));
}

// leaf_hdr: every value is preceded by its 5-byte byte-tree leaf header (checked on the device; *format_ok)
// (checked_on_host: the caller has validated the values itself -- the conversion is only queued, no verdict is read back)
static int import_be(vmn_ctx* ctx, const vmn_modulus& m, size_t nbytes, const uint8_t* be, size_t n, uint32_t* d_out,
                     int* all_in_range, int leaf_hdr = 0, int* format_ok = nullptr, bool checked_on_host = false) {
    if (all_in_range) *all_in_range = 1;
    if (format_ok) *format_ok = 1;
    if (n == 0) return VMN_OK;
    // a curve point in a byte tree is node(leaf(x), leaf(y)): 15 framing bytes around the two coordinates
    const size_t stride = m.ec ? 2 * nbytes + (leaf_hdr ? 15 : 0) : nbytes + (leaf_hdr ? 5 : 0);
    DevTmp raw(ctx);
    VMN_TRY(raw.alloc(n * stride + 8));
    if (checked_on_host || n * stride <= vmn_ctx::UP_SLOT) VMN_TRY(h2d(ctx, raw.p, be, n * stride));    // (small: a pinned slot, queued)
    else VMN_HIP(hipMemcpyAsync(raw.p, be, n * stride, hipMemcpyHostToDevice, ctx->stream));
    VMN_TRY(dev_zero(ctx, ctx->flags, sizeof(uint32_t)));
    note_work(ctx, m, (m.ec ? 7.0 : 1.0) * (double)n);
    int rc = VMN_ERR_ARG;
    if (m.ec) {
#define X(S_, NW_)                                                                                                   \
    if (m.ec->S == S_)                                                                                               \
        rc = launch_light(ctx, "import", k_ec_import<S_, NW_>, grid_for(n), d_out, (const uint8_t*)raw.as<uint8_t>(), \
                          nbytes, stride, leaf_hdr, n, ecdev(m.ec), ctx->flags);
        VMN_FOR_CURVES(X)
#undef X
    } else {
#define X(S_, NW_, LPE_)                                                                                              \
    if (m.S == S_)                                                                                              \
        rc = launch(ctx, "import", k_import_be<Cfg<S_, LPE_>, NW_>, egrid(m, n), lds_bytes(m), d_out, raw.as<uint8_t>(), \
                    nbytes, stride, leaf_hdr, n, m.d_n, m.n0inv, m.d_rr, ctx->flags);
    VMN_DISPATCH(n, X)
#undef X
    }
    VMN_TRY(rc);
    if (checked_on_host) return VMN_OK;
    uint32_t fl = 0;
    VMN_TRY(read_flag(ctx, &fl));
    if (all_in_range) *all_in_range = (fl & 1u) ? 0 : 1;
    if (format_ok) *format_ok = (fl & 4u) ? 0 : 1;
    return VMN_OK;
}

// A handful of curve points (the single elements a proof reads back: a product, the last B, h_0): the Jacobian rows come to the
// host as they are and the host makes them affine (one binary-Euclid inversion per point, ~10 us) -- the device's export does a
// Fermat inversion per point, a chain of ~380 dependent field products on one lane: 155 us of latency whatever the count
// (profiles/r04_timeline_p256_n10000.txt), and a proof over a curve reads single points a dozen times.
static int ec_export_few_host(vmn_ctx* ctx, const vmn_modulus& m, size_t nbytes, const uint32_t* d_in, size_t n, uint8_t* be) {
    const int S = m.ec->S;
    const size_t FW = (size_t)stride_for_limbs(S), ROW = 3 * FW;
    std::vector<uint32_t> rows(n * ROW);
    VMN_TRY(d2h(ctx, rows.data(), d_in, n * ROW * sizeof(uint32_t)));
    const num64::Mod& F = *m.ec->f64;
    // values are x R_d mod p with R_d = 2^(28 S), limbs normalised but the value lazy (a small multiple of p above x R_d)
    const num64::Num& rdinv = m.ec->rd_inv;
    auto field = [&](const uint32_t* limbs) {
        uint8_t buf[96] = {0};                                  // big-endian: up to 21 limbs of 28 bits and a 32-bit top limb
        const size_t nb = sizeof(buf);
        unsigned __int128 acc = 0;                              // the bits not yet written, the lowest at the next byte position
        int acc_bits = 0;
        size_t pos = nb;
        for (int j = 0; j < S; ++j) {                           // limb j sits at bit 28 j (a limb wider than 28 bits just carries on)
            acc += (unsigned __int128)limbs[j] << acc_bits;
            acc_bits += 28;
            while (acc_bits >= 8) {
                buf[--pos] = (uint8_t)acc;
                acc >>= 8;
                acc_bits -= 8;
            }
        }
        while (acc != 0) {
            buf[--pos] = (uint8_t)acc;
            acc >>= 8;
        }
        return F.mul(F.reduce(buf, nb), rdinv);
    };
    for (size_t i = 0; i < n; ++i) {
        const uint32_t* row = rows.data() + i * ROW;
        uint8_t* dst = be + i * 2 * nbytes;
        if (row[ROW - 1]) {                                     // the point at infinity: both coordinates all 0xff
            memset(dst, 0xff, 2 * nbytes);
            continue;
        }
        const num64::Num X = field(row), Y = field(row + FW), Z = field(row + 2 * FW);
        const num64::Num zi = F.inv(Z), zi2 = F.mul(zi, zi);
        num64::to_be(F.mul(X, zi2), dst, nbytes);
        num64::to_be(F.mul(Y, F.mul(zi2, zi)), dst + nbytes, nbytes);
    }
    return VMN_OK;
}

// (pinned_async: `be` is pinned host memory and the copy is only queued -- the caller orders itself behind it)
static int ec_normalize(vmn_ctx* ctx, const vmn_modulus& m, const uint32_t* const* ins, size_t k, size_t n, uint32_t* out);
static int export_be(vmn_ctx* ctx, const vmn_modulus& m, size_t nbytes, const uint32_t* d_in, size_t n, uint8_t* be,
                     int leaf_hdr = 0, bool pinned_async = false) {
    if (n == 0) return VMN_OK;
    // a curve point in a byte tree is node(leaf(x), leaf(y)): 15 framing bytes around the two coordinates
    const size_t stride = m.ec ? 2 * nbytes + (leaf_hdr ? 15 : 0) : nbytes + (leaf_hdr ? 5 : 0);
    if (m.ec && n <= 4 && !leaf_hdr && !pinned_async && !getenv("VMN_EC_EXPORT_DEVICE")) return ec_export_few_host(ctx, m, nbytes, d_in, n, be);
    DevTmp raw(ctx);
    VMN_TRY(raw.alloc(n * stride + 8));
    // Curves: the export kernel inverts Z per point -- a Fermat power of ~380 products, 25 times the rest of a point's export.
    // One chain of them lasts 0.16 ms whatever the array (up to a device full of lanes); from there on the rows are normalised
    // first by the batched inversion of the multi-exponentiations (~8 products per point) and exported as they are.
    DevTmp affine_rows(ctx);
    int ec_rows_affine = 0;
    if (m.ec) {
        const char* env = getenv("VMN_EC_EXPORT_NORMALISE_MIN");          // (read per call: the tests run both exports at their sizes)
        const size_t min_n = env && *env ? (size_t)strtoull(env, nullptr, 10) : (size_t)262144;
        if (n >= min_n) {
            VMN_TRY(affine_rows.alloc(n * (size_t)m.W * sizeof(uint32_t)));
            const uint32_t* one_array[1] = {d_in};
            VMN_TRY(ec_normalize(ctx, m, one_array, 1, n, affine_rows.as<uint32_t>()));
            d_in = affine_rows.as<uint32_t>();
            ec_rows_affine = 2;
        }
    }
    if (ec_rows_affine) note_work(ctx, m, 2.0 * (double)n);
    else note_work(ctx, m, m.ec ? 8.0 * (double)n : (double)n, m.ec ? (double)m.nbits * (double)n : 0.0);     // curves: one Fermat inversion per point
    int rc = VMN_ERR_ARG;
    if (m.ec) {
#define X(S_, NW_)                                                                                               \
    if (m.ec->S == S_)                                                                                           \
        rc = launch_light(ctx, "export", k_ec_export<S_, NW_>, grid_for(n), raw.as<uint8_t>(), nbytes, stride,   \
                          leaf_hdr | ec_rows_affine, d_in, n, ecdev(m.ec));
        VMN_FOR_CURVES(X)
#undef X
    } else {
#define X(S_, NW_, LPE_)                                                                                          \
    if (m.S == S_)                                                                                          \
        rc = launch(ctx, "export", k_export_be<Cfg<S_, LPE_>, NW_>, egrid(m, n), lds_bytes(m), raw.as<uint8_t>(),    \
                    nbytes, stride, leaf_hdr, d_in, n, m.d_n, m.n0inv);
    VMN_DISPATCH(n, X)
#undef X
    }
    VMN_TRY(rc);
    if (pinned_async) {
        VMN_HIP(hipMemcpyAsync(be, raw.p, n * stride, hipMemcpyDeviceToHost, ctx->stream));    // (raw: reuse is ordered on this stream)
        return VMN_OK;
    }
    return d2h(ctx, be, raw.p, n * stride);
}

// byte tree of an array: node header on the host, leaves framed on the device
// node(N leaves of nbytes) for residues; node(N x node(leaf(x), leaf(y))) for curve points (nbytes = one coordinate)
static size_t bytetree_size(size_t n, size_t nbytes, bool ec = false) { return 5 + n * (ec ? 15 + 2 * nbytes : 5 + nbytes); }
static int to_bytetree(vmn_ctx* ctx, const vmn_modulus& m, size_t nbytes, const uint32_t* d_in, size_t n, uint8_t* out) {
    out[0] = 0;
    out[1] = (uint8_t)(n >> 24);
    out[2] = (uint8_t)(n >> 16);
    out[3] = (uint8_t)(n >> 8);
    out[4] = (uint8_t)n;
    return export_be(ctx, m, nbytes, d_in, n, out + 5, 1);
}
// returns the element count through *n_out; *format_ok = 0 if the buffer is not node(N leaves of nbytes)
static int bytetree_header(const uint8_t* bt, size_t len, size_t nbytes, size_t expected_n, size_t* n_out, int* format_ok, bool ec = false) {
    *format_ok = 0;
    *n_out = 0;
    if (len < 5 || bt[0] != 0) return VMN_OK;
    size_t n = ((size_t)bt[1] << 24) | ((size_t)bt[2] << 16) | ((size_t)bt[3] << 8) | bt[4];
    if (len != bytetree_size(n, nbytes, ec)) return VMN_OK;
    if (expected_n != 0 && n != expected_n) return VMN_OK;
    *n_out = n;
    *format_ok = 1;
    return VMN_OK;
}

// one element (big-endian) -> device, M28 form
// (residues and scalars: "0 <= value < modulus" is decided on the host, so the conversion is queued without a read-back --
// a pushed element or a scalar operand in front of a kernel no longer drains the stream; curve points keep the device's
// on-curve verdict)
static int import_one(vmn_ctx* ctx, const vmn_modulus& m, size_t nbytes, const uint8_t* be, uint32_t** d_out) {
    bool host_checked = false;
    if (!m.ec) {
        bool in_range = true;
        for (size_t i = 0; i + 4 * (size_t)m.NW < nbytes; ++i) in_range = in_range && be[i] == 0;       // bytes above the packed words
        in_range = in_range && hostbig::cmp(hostbig::from_be(be, nbytes, m.NW), m.n_words) < 0;
        if (!in_range) {
            set_error("scalar operand out of range");
            *d_out = nullptr;
            return VMN_ERR_FORMAT;
        }
        host_checked = true;
    }
    VMN_TRY(alloc_elems(ctx, m, 1, d_out));
    int ok = 1;
    int rc = import_be(ctx, m, nbytes, be, 1, *d_out, &ok, 0, nullptr, host_checked);
    if (rc == VMN_OK && !ok) {
        set_error("scalar operand out of range");
        rc = VMN_ERR_FORMAT;
    }
    if (rc != VMN_OK) {
        pool_free(ctx, *d_out, elems_bytes(m, 1));
        *d_out = nullptr;
    }
    return rc;
}
static void free_one(vmn_ctx* ctx, const vmn_modulus& m, uint32_t* d) { pool_free(ctx, d, elems_bytes(m, 1)); }

static int mul_arrays(vmn_ctx* ctx, const vmn_modulus& m, const uint32_t* x, const uint32_t* y, size_t ystride, size_t n,
                      uint32_t* out) {
    if (n == 0) return VMN_OK;
    note_work(ctx, m, (m.ec ? EC_ADD : 1.0) * (double)n);
    int rc = VMN_ERR_ARG;
    if (m.ec) {
#define X(S_, NW_) \
    if (m.ec->S == S_) rc = launch_light(ctx, "modmul", k_ec_add<S_>, grid_for(n), out, x, y, ystride, n, ecdev(m.ec));
        VMN_FOR_CURVES(X)
#undef X
        return rc;
    }
#define X(S_, NW_, LPE_) \
    if (m.S == S_) rc = launch(ctx, "modmul", k_mul<Cfg<S_, LPE_>>, egrid(m, n), lds_bytes(m), out, x, y, ystride, n, m.d_n, m.n0inv);
    VMN_DISPATCH(n, X)
#undef X
    return rc;
}

// M28 residues -> packed words (n * NW words)
static int to_words(vmn_ctx* ctx, const vmn_modulus& m, const uint32_t* in, size_t n, uint32_t* out_words) {
    if (n == 0) return VMN_OK;
    note_work(ctx, m, (double)n);
    int rc = VMN_ERR_ARG;
#define X(S_, NW_, LPE_) \
    if (m.S == S_) rc = launch(ctx, "to_words", k_to_words<Cfg<S_, LPE_>, NW_>, egrid(m, n), lds_bytes(m), out_words, in, n, m.d_n, m.n0inv);
    VMN_DISPATCH(n, X)
#undef X
    return rc;
}

// window size minimising (table build) + (window multiplications)
static int pick_window(int ebits) {
    int best = 1;
    long best_cost = 1L << 60;
    for (int w = 1; w <= 7; ++w) {
        long cost = (1L << w) - 2 + (ebits + w - 1) / w;
        if (cost < best_cost) {
            best_cost = cost;
            best = w;
        }
    }
    return best;
}

// out[i] = x[i]^e[i] with packed-word exponents already on the device
static int modpow_words(vmn_ctx* ctx, const vmn_modulus& m0, const uint32_t* x, const uint32_t* e_words, int ewords,
                        size_t estride, int ebits, size_t n, uint32_t* out) {
    if (n == 0) return VMN_OK;
    if (ebits < 1) ebits = 1;
    if (m0.ec) {
        const vmn_modulus& m = m0;
        int wb = std::min(pick_window(ebits), 5);
        unsigned grid = std::min<unsigned>(grid_for(n), (unsigned)(ctx->num_cus * 2));
        size_t tab_bytes = (size_t)grid * BLOCK * ((size_t)1 << wb) * elem_words(m) * sizeof(uint32_t);
        VMN_TRY(ensure_scratch(ctx, tab_bytes));
        note_work(ctx, m, (double)n * (EC_DBL * ebits + EC_ADD * ((ebits + wb - 1) / wb + (1 << wb))));
        int rc = VMN_ERR_ARG;
#define X(S_, NW_)                                                                                                  \
    if (m.ec->S == S_)                                                                                              \
        rc = launch_light(ctx, "modpow", k_ec_mulvar<S_>, grid, out, x, e_words, ewords, estride, ebits, wb, n,      \
                          ecdev(m.ec), reinterpret_cast<uint32_t*>(ctx->scratch));
        VMN_FOR_CURVES(X)
#undef X
        return rc;
    }
    int wbits = pick_window(ebits);
    const vmn_modulus& m = geom(ctx, m0, n);
    unsigned max_blocks = (unsigned)(ctx->num_cus * blocks_per_cu(m));
    unsigned grid = std::min<unsigned>(egrid(m, n), max_blocks);
    size_t tab_bytes = (size_t)grid * (BLOCK / m.LPE) * ((size_t)1 << wbits) * elem_words(m) * sizeof(uint32_t);
    VMN_TRY(ensure_scratch(ctx, tab_bytes));
    {
        const int nwin = (ebits + wbits - 1) / wbits;
        note_work(ctx, m, (double)n * (nwin - 1 + (1 << wbits) - 2), (double)n * (nwin - 1) * wbits);
    }
    int rc = VMN_ERR_ARG;
    // More than one round of tiles: the power in phases from a queue of (phase, tile) units (k_modpow_phased), so that the
    // launch ends with a tail of one unit instead of one whole tile -- 10^6 elements are 7.63 rounds and cost 8 otherwise.
    // VMN_MODPOW_PHASES: 1 = one phase (k_modpow); default 16.
    static const int phases_env = [] {
        const char* e = getenv("VMN_MODPOW_PHASES");
        return e && *e ? std::max(1, atoi(e)) : 16;
    }();
    if (const char* mb = getenv("VMN_MODPOW_MAX_BLOCKS")) {       // test hook: a "device" of few workgroup slots, so that small arrays take the phased kernel
        const int v = atoi(mb);
        if (v >= 1) max_blocks = std::min<unsigned>(max_blocks, (unsigned)v);
    }
    const size_t epb = (size_t)(BLOCK / m.LPE);
    const size_t ntiles = (n + epb - 1) / epb;
    const int main_windows = (ebits + wbits - 1) / wbits - 1;
    const int phases = ntiles > (size_t)max_blocks && ntiles < ((size_t)1 << 26) ? std::min(phases_env, std::max(1, main_windows)) : 1;
    // a window table per ELEMENT (9.5 GB for 10^6 x 2048 bits, w = 5): where that does not fit -- more than 64 GB, or the
    // allocation fails -- the power runs tile by tile with a table per lane slot, as it always did
    const size_t ptab_bytes = ntiles * epb * ((size_t)1 << wbits) * elem_words(m) * sizeof(uint32_t);
    DevTmp ptab(ctx), sync_words(ctx);
    bool phased = phases > 1 && ptab_bytes <= ((size_t)64 << 30);
    if (phased && ptab.alloc(ptab_bytes) != VMN_OK) phased = false;
    if (phased) {
        VMN_TRY(sync_words.alloc((ntiles + 1) * sizeof(uint32_t)));
        VMN_TRY(dev_zero(ctx, sync_words.p, (ntiles + 1) * sizeof(uint32_t)));
        uint32_t* queue = sync_words.as<uint32_t>();
#define X(S_, NW_, LPE_)                                                                                                 \
    if (m.S == S_)                                                                                                 \
        rc = launch(ctx, "modpow", k_modpow_phased<Cfg<S_, LPE_>>, max_blocks, lds_bytes(m), out, x, e_words, ewords, estride, ebits, \
                    wbits, n, m.d_n, m.n0inv, m.d_one, ptab.as<uint32_t>(), phases, queue, queue + 1);
        VMN_FOR_SIZES(X)
#undef X
        return rc;
    }
#define X(S_, NW_, LPE_)                                                                                                 \
    if (m.S == S_)                                                                                                 \
        rc = launch(ctx, "modpow", k_modpow<Cfg<S_, LPE_>>, grid, lds_bytes(m), out, x, e_words, ewords, estride, ebits, wbits, \
                    n, m.d_n, m.n0inv, m.d_one, reinterpret_cast<uint32_t*>(ctx->scratch));
    VMN_FOR_SIZES(X)
#undef X
    return rc;
}

// out[i] = x[i]^e for ONE exponent: left-to-right sliding window (csrc/modp_shared_exp.h).  Modular groups, exponents of
// 33 bits and more (shorter ones, zero included, take the fixed-window kernel).
static int sliding_window_bits(int ebits) {
    if (const char* env = getenv("VMN_SLIDING_WINDOW")) {          // measurement knob; 0 = use the fixed window
        int w = atoi(env);
        if (w >= 0 && w <= 8) return w;
    }
    int best = 2;
    long best_cost = 1L << 60;
    for (int w = 2; w <= 8; ++w) {
        long cost = (1L << (w - 1)) + ebits / (w + 1);
        if (cost < best_cost) {
            best_cost = cost;
            best = w;
        }
    }
    return best;
}
static int modpow_shared(vmn_ctx* ctx, const vmn_modulus& m0, const uint32_t* x, const Big& e, int ebits, size_t n, uint32_t* out) {
    const int w = sliding_window_bits(ebits);
    std::vector<SlideStep> steps;
    int pending = 0;
    long mults = 0, squarings = 0;
    for (int i = ebits - 1; i >= 0;) {
        if (!hostbig::get_bit(e, i)) {
            ++pending;
            --i;
            continue;
        }
        int l = std::max(i - w + 1, 0);
        while (!hostbig::get_bit(e, l)) ++l;                      // the window ends in a one: its value is odd
        uint32_t val = 0;
        for (int b = i; b >= l; --b) val = (val << 1) | (uint32_t)hostbig::get_bit(e, b);
        steps.push_back(SlideStep{steps.empty() ? 0 : pending + (i - l + 1), (int)((val - 1) / 2)});
        pending = 0;
        i = l - 1;
    }
    if (pending) steps.push_back(SlideStep{pending, -1});
    for (size_t k = 1; k < steps.size(); ++k) {
        squarings += steps[k].sq;
        mults += steps[k].idx >= 0 ? 1 : 0;
    }
    const int tsize = 1 << (w - 1);
    const vmn_modulus& m = geom(ctx, m0, n);
    unsigned max_blocks = (unsigned)(ctx->num_cus * blocks_per_cu(m));
    if (const char* mb = getenv("VMN_MODPOW_MAX_BLOCKS")) {       // (test hook, see modpow_words)
        const int v = atoi(mb);
        if (v >= 1) max_blocks = std::min<unsigned>(max_blocks, (unsigned)v);
    }
    const unsigned grid = std::min<unsigned>(egrid(m, n), max_blocks);
    const size_t tab_bytes = (size_t)grid * (BLOCK / m.LPE) * (size_t)tsize * elem_words(m) * sizeof(uint32_t);
    VMN_TRY(ensure_scratch(ctx, tab_bytes));
    DevTmp dsteps(ctx);
    VMN_TRY(dsteps.alloc(steps.size() * sizeof(SlideStep)));
    VMN_TRY(h2d(ctx, dsteps.p, steps.data(), steps.size() * sizeof(SlideStep)));
    note_work(ctx, m, (double)n * (double)(mults + tsize - 1), (double)n * (double)(squarings + 1));
    int rc = VMN_ERR_ARG;
    // more than one round of tiles: in phases from a queue of units (k_modpow_shared_phased; modpow_words has the reasons)
    {
        const size_t epb = (size_t)(BLOCK / m.LPE), ntiles = (n + epb - 1) / epb;
        static const int phases_env3 = [] {
            const char* e = getenv("VMN_MODPOW_PHASES");
            return e && *e ? std::max(1, atoi(e)) : 16;
        }();
        const int phases = ntiles > (size_t)max_blocks && ntiles < ((size_t)1 << 26) ? std::min(phases_env3, std::max(1, (int)steps.size() - 1)) : 1;
        const size_t ptab_bytes = ntiles * epb * (size_t)tsize * elem_words(m) * sizeof(uint32_t);
        DevTmp ptab(ctx), sync_words(ctx);
        bool phased = phases > 1 && ptab_bytes <= ((size_t)64 << 30);
        if (phased && ptab.alloc(ptab_bytes) != VMN_OK) phased = false;
        if (phased) {
            VMN_TRY(sync_words.alloc((ntiles + 1) * sizeof(uint32_t)));
            VMN_TRY(dev_zero(ctx, sync_words.p, (ntiles + 1) * sizeof(uint32_t)));
            uint32_t* queue = sync_words.as<uint32_t>();
#define X(S_, NW_, LPE_)                                                                                                 \
    if (m.S == S_)                                                                                                 \
        rc = launch(ctx, "modpow", k_modpow_shared_phased<Cfg<S_, LPE_>>, max_blocks, lds_bytes(m), out, x,                  \
                    (const SlideStep*)dsteps.as<SlideStep>(), (int)steps.size(), tsize, n, m.d_n, m.n0inv, ptab.as<uint32_t>(), \
                    phases, queue, queue + 1);
            VMN_FOR_SIZES(X)
#undef X
            return rc;
        }
    }
#define X(S_, NW_, LPE_)                                                                                                 \
    if (m.S == S_)                                                                                                 \
        rc = launch(ctx, "modpow", k_modpow_shared<Cfg<S_, LPE_>>, grid, lds_bytes(m), out, x, (const SlideStep*)dsteps.as<SlideStep>(), \
                    (int)steps.size(), tsize, n, m.d_n, m.n0inv, reinterpret_cast<uint32_t*>(ctx->scratch));
    VMN_FOR_SIZES(X)
#undef X
    return rc;
}

// big-endian integers (ebytes each) -> packed little-endian words on the host
static void be_ints_to_words(const uint8_t* be, size_t ebytes, size_t n, int ewords, std::vector<uint32_t>& out) {
    out.assign(n * (size_t)ewords, 0);
    for (size_t i = 0; i < n; ++i) {
        const uint8_t* p = be + i * ebytes;
        uint32_t* w = out.data() + i * ewords;
        for (size_t b = 0; b < ebytes; ++b) {
            size_t k = ebytes - 1 - b;
            if (k / 4 < (size_t)ewords) w[k / 4] |= (uint32_t)p[b] << (8 * (k % 4));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// group element arrays
// ------------------------------------------------------------------------------------------------
static int new_garray(vmn_group* grp, size_t n, vmn_garray** out) {
    std::unique_ptr<vmn_garray> a(new vmn_garray());
    a->grp = grp;
    a->n = n;
    a->bytes = elems_bytes(grp->P, n);
    a->lane = LANE(grp->ctx);
    VMN_TRY(alloc_elems(a->lane, grp->P, n, &a->d));
    *out = a.release();
    return VMN_OK;
}
static int new_rarray(vmn_group* grp, size_t n, vmn_rarray** out) {
    std::unique_ptr<vmn_rarray> a(new vmn_rarray());
    a->grp = grp;
    a->n = n;
    a->bytes = elems_bytes(grp->Q, n);
    a->lane = LANE(grp->ctx);
    VMN_TRY(alloc_elems(a->lane, grp->Q, n, &a->d));
    *out = a.release();
    return VMN_OK;
}

// A block goes back to the pool of the lane it was allocated on -- the pool's one guarantee is "reuse is ordered on that
// lane's stream".  When another lane's thread frees it (the helper dropping the last reference to a main-lane array, or
// the other way round), whatever that thread has queued on ITS stream may still touch the block: the owner's stream is
// made to wait for it (an event, no host blocking) before the block can be handed out again.  Work queued on a third
// party's behalf is the caller's to order: an array must outlive the calls that use it, as with any library.
static void free_to_owner(vmn_ctx* owner, vmn_ctx* caller, void* d, size_t bytes) {
    if (!owner) owner = caller;
    if (caller != owner && d) {
        hipEvent_t ev = nullptr;
        bool ordered = false;
        if (hipSetDevice(caller->device) == hipSuccess && hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
            ordered = hipEventRecord(ev, caller->stream) == hipSuccess && hipStreamWaitEvent(owner->stream, ev, 0) == hipSuccess;
            (void)hipEventDestroy(ev);                 // released by the runtime once the wait has been satisfied
        }
        if (!ordered) (void)hipStreamSynchronize(caller->stream);
    }
    std::lock_guard<std::recursive_mutex> guard__(owner->mu);
    pool_free(owner, d, bytes);
}
extern "C" void vmn_garray_free(vmn_garray* a) {
    if (!a) return;
    free_to_owner(a->lane, LANE(a->grp->ctx), a->d, a->bytes);
    delete a;
}
extern "C" void vmn_rarray_free(vmn_rarray* a) {
    if (!a) return;
    free_to_owner(a->lane, LANE(a->grp->ctx), a->d, a->bytes);
    delete a;
}
extern "C" vmn_group* vmn_garray_group(const vmn_garray* a) { return a ? a->grp : nullptr; }
extern "C" vmn_group* vmn_rarray_group(const vmn_rarray* a) { return a ? a->grp : nullptr; }
extern "C" size_t vmn_garray_size(const vmn_garray* a) { return a ? a->n : 0; }
extern "C" size_t vmn_rarray_size(const vmn_rarray* a) { return a ? a->n : 0; }

extern "C" int vmn_garray_from_be(vmn_group* grp, const uint8_t* be, size_t n, vmn_garray** out, int* all_in_range) {
    ARG_CHECK(grp && out && (be || n == 0), "null argument");
    VMN_ENTER(LANE(grp->ctx));
    vmn_garray* a = nullptr;
    VMN_TRY(new_garray(grp, n, &a));
    int rc = import_be(LANE(grp->ctx), grp->P, grp->nbytes, be, n, a->d, all_in_range);
    if (rc != VMN_OK) {
        vmn_garray_free(a);
        return rc;
    }
    *out = a;
    return VMN_OK;
}
extern "C" int vmn_rarray_from_be(vmn_group* grp, const uint8_t* be, size_t n, vmn_rarray** out, int* all_in_range) {
    ARG_CHECK(grp && out && (be || n == 0), "null argument");
    VMN_ENTER(LANE(grp->ctx));
    vmn_rarray* a = nullptr;
    VMN_TRY(new_rarray(grp, n, &a));
    int rc = import_be(LANE(grp->ctx), grp->Q, grp->xbytes, be, n, a->d, all_in_range);
    if (rc != VMN_OK) {
        vmn_rarray_free(a);
        return rc;
    }
    *out = a;
    return VMN_OK;
}
extern "C" size_t vmn_garray_bytetree_size(const vmn_garray* a) { return a ? bytetree_size(a->n, a->grp->nbytes, a->grp->curve != nullptr) : 0; }
extern "C" size_t vmn_rarray_bytetree_size(const vmn_rarray* a) { return a ? bytetree_size(a->n, a->grp->xbytes) : 0; }
extern "C" int vmn_garray_to_bytetree(const vmn_garray* a, uint8_t* out) {
    ARG_CHECK(a && out, "null argument");
    VMN_ENTER(LANE(a->grp->ctx));
    return to_bytetree(LANE(a->grp->ctx), a->grp->P, a->grp->nbytes, a->d, a->n, out);
}
extern "C" int vmn_rarray_to_bytetree(const vmn_rarray* a, uint8_t* out) {
    ARG_CHECK(a && out, "null argument");
    VMN_ENTER(LANE(a->grp->ctx));
    return to_bytetree(LANE(a->grp->ctx), a->grp->Q, a->grp->xbytes, a->d, a->n, out);
}
extern "C" int vmn_garray_from_bytetree(vmn_group* grp, const uint8_t* bt, size_t len, size_t expected_n, vmn_garray** out,
                                        int* format_ok, int* all_in_range) {
    ARG_CHECK(grp && bt && out && format_ok, "null argument");
    VMN_ENTER(LANE(grp->ctx));
    *out = nullptr;
    size_t n = 0;
    VMN_TRY(bytetree_header(bt, len, grp->nbytes, expected_n, &n, format_ok, grp->curve != nullptr));
    if (!*format_ok) return VMN_OK;
    vmn_garray* a = nullptr;
    VMN_TRY(new_garray(grp, n, &a));
    int rc = import_be(LANE(grp->ctx), grp->P, grp->nbytes, bt + 5, n, a->d, all_in_range, 1, format_ok);
    if (rc != VMN_OK || !*format_ok) {
        vmn_garray_free(a);
        return rc;
    }
    *out = a;
    return VMN_OK;
}
extern "C" int vmn_rarray_from_bytetree(vmn_group* grp, const uint8_t* bt, size_t len, size_t expected_n, vmn_rarray** out,
                                        int* format_ok, int* all_in_range) {
    ARG_CHECK(grp && bt && out && format_ok, "null argument");
    VMN_ENTER(LANE(grp->ctx));
    *out = nullptr;
    size_t n = 0;
    VMN_TRY(bytetree_header(bt, len, grp->xbytes, expected_n, &n, format_ok));
    if (!*format_ok) return VMN_OK;
    vmn_rarray* a = nullptr;
    VMN_TRY(new_rarray(grp, n, &a));
    int rc = import_be(LANE(grp->ctx), grp->Q, grp->xbytes, bt + 5, n, a->d, all_in_range, 1, format_ok);
    if (rc != VMN_OK || !*format_ok) {
        vmn_rarray_free(a);
        return rc;
    }
    *out = a;
    return VMN_OK;
}
extern "C" int vmn_garray_to_be(const vmn_garray* a, uint8_t* be_out) {
    ARG_CHECK(a && (be_out || a->n == 0), "null argument");
    VMN_ENTER(LANE(a->grp->ctx));
    return export_be(LANE(a->grp->ctx), a->grp->P, a->grp->nbytes, a->d, a->n, be_out);
}
extern "C" int vmn_rarray_to_be(const vmn_rarray* a, uint8_t* be_out) {
    ARG_CHECK(a && (be_out || a->n == 0), "null argument");
    VMN_ENTER(LANE(a->grp->ctx));
    return export_be(LANE(a->grp->ctx), a->grp->Q, a->grp->xbytes, a->d, a->n, be_out);
}

extern "C" int vmn_garray_mul(const vmn_garray* x, const vmn_garray* y, vmn_garray** out) {
    ARG_CHECK(x && y && out, "null argument");
    ARG_CHECK(x->grp == y->grp && x->n == y->n, "arrays differ in group or size");
    vmn_group* g = x->grp;
    VMN_ENTER(LANE(g->ctx));
    vmn_garray* r = nullptr;
    VMN_TRY(new_garray(g, x->n, &r));
    int rc = mul_arrays(LANE(g->ctx), g->P, x->d, y->d, elem_words(g->P), x->n, r->d);
    if (rc != VMN_OK) {
        vmn_garray_free(r);
        return rc;
    }
    *out = r;
    return VMN_OK;
}

extern "C" int vmn_garray_exp_array(const vmn_garray* x, const vmn_rarray* e, int ebits, vmn_garray** out) {
    ARG_CHECK(x && e && out, "null argument");
    ARG_CHECK(x->grp == e->grp && x->n == e->n, "arrays differ in group or size");
    vmn_group* g = x->grp;
    vmn_ctx* ctx = LANE(g->ctx);
    VMN_ENTER(ctx);
    if (ebits <= 0 || ebits > g->Q.nbits) ebits = g->Q.nbits;
    vmn_garray* r = nullptr;
    VMN_TRY(new_garray(g, x->n, &r));
    DevTmp ew(ctx);
    int rc = ew.alloc(x->n * (size_t)g->Q.NW * sizeof(uint32_t));
    if (rc == VMN_OK) rc = to_words(ctx, g->Q, e->d, e->n, ew.as<uint32_t>());
    if (rc == VMN_OK) rc = modpow_words(ctx, g->P, x->d, ew.as<uint32_t>(), g->Q.NW, (size_t)g->Q.NW, ebits, x->n, r->d);
    if (rc != VMN_OK) {
        vmn_garray_free(r);
        return rc;
    }
    *out = r;
    return VMN_OK;
}

extern "C" int vmn_garray_exp_ints(const vmn_garray* x, const uint8_t* exps_be, size_t ebytes, int ebits, vmn_garray** out) {
    ARG_CHECK(x && out && (exps_be || x->n == 0) && ebytes > 0, "null argument");
    vmn_group* g = x->grp;
    vmn_ctx* ctx = LANE(g->ctx);
    VMN_ENTER(ctx);
    if (ebits <= 0 || (size_t)ebits > 8 * ebytes) ebits = (int)(8 * ebytes);
    int ewords = (ebits + 31) / 32;
    std::vector<uint32_t> hw;
    be_ints_to_words(exps_be, ebytes, x->n, ewords, hw);
    vmn_garray* r = nullptr;
    VMN_TRY(new_garray(g, x->n, &r));
    DevTmp ew(ctx);
    int rc = ew.alloc(hw.size() * sizeof(uint32_t));
    if (rc == VMN_OK && !hw.empty()) {
        if (hw.size() * sizeof(uint32_t) <= vmn_ctx::UP_SLOT) {
            rc = h2d(ctx, ew.p, hw.data(), hw.size() * sizeof(uint32_t));      // (a pinned slot: queued, no synchronisation)
        } else {
            hipError_t he = hipMemcpyAsync(ew.p, hw.data(), hw.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
            if (he == hipSuccess) he = hipStreamSynchronize(ctx->stream);   // hw is pageable host memory
            if (he != hipSuccess) {
                set_error("exponent upload failed: %s", hipGetErrorString(he));
                rc = VMN_ERR_DEVICE;
            }
        }
    }
    if (rc == VMN_OK) rc = modpow_words(ctx, g->P, x->d, ew.as<uint32_t>(), ewords, (size_t)ewords, ebits, x->n, r->d);
    if (rc != VMN_OK) {
        vmn_garray_free(r);
        return rc;
    }
    *out = r;
    return VMN_OK;
}

extern "C" int vmn_garray_exp_scalar(const vmn_garray* x, const uint8_t* e_be, size_t ebytes, vmn_garray** out) {
    ARG_CHECK(x && e_be && out && ebytes > 0, "null argument");
    vmn_group* g = x->grp;
    vmn_ctx* ctx = LANE(g->ctx);
    VMN_ENTER(ctx);
    int ewords = (int)((ebytes + 3) / 4);
    Big e = hostbig::from_be(e_be, ebytes, ewords);
    int ebits = std::max(1, hostbig::bit_length(e));
    ewords = (ebits + 31) / 32;
    vmn_garray* r = nullptr;
    VMN_TRY(new_garray(g, x->n, &r));
    DevTmp ew(ctx);
    int rc = ew.alloc(ewords * sizeof(uint32_t));
    if (rc == VMN_OK) {
        rc = h2d(ctx, ew.p, e.data(), ewords * sizeof(uint32_t));
    }
    if (rc == VMN_OK && !g->P.ec && ebits > 32 && x->n > 0 && sliding_window_bits(ebits) > 0)
        rc = modpow_shared(ctx, g->P, x->d, e, ebits, x->n, r->d);                 // one exponent for all: sliding window
    else if (rc == VMN_OK)
        rc = modpow_words(ctx, g->P, x->d, ew.as<uint32_t>(), ewords, 0, ebits, x->n, r->d);
    if (rc != VMN_OK) {
        vmn_garray_free(r);
        return rc;
    }
    *out = r;
    return VMN_OK;
}

// out[i] = x[i]^e * y[i]^f[i]: one simultaneous power (k_modpow2), the squarings shared between the two exponents.
// The verifiers' check (B): B_i^v (B_{i-1}^{-1})^{k_E,i}.  Modular groups only.
extern "C" int vmn_garray_exp2(const vmn_garray* x, const uint8_t* e_be, size_t ebytes, const vmn_garray* y, const vmn_rarray* f,
                               int fbits, vmn_garray** out) {
    ARG_CHECK(x && e_be && ebytes > 0 && y && f && out, "null argument");
    ARG_CHECK(x->grp == y->grp && x->grp == f->grp && x->n == y->n && x->n == f->n, "arrays differ in group or size");
    vmn_group* g = x->grp;
    vmn_ctx* ctx = LANE(g->ctx);
    VMN_ENTER(ctx);
    if (fbits <= 0 || fbits > g->Q.nbits) fbits = g->Q.nbits;
    const size_t n = x->n;
    int ewords = (int)((ebytes + 3) / 4);
    Big e = hostbig::from_be(e_be, ebytes, ewords);
    if (g->P.ec) {                                       // scalars act modulo the group order
        const vmn_modulus& q = g->Q;
        std::vector<uint8_t> be((size_t)q.NW * 4);
        num64::to_be(q.hm64->reduce(e_be, ebytes), be.data(), be.size());
        e = hostbig::from_be(be.data(), be.size(), q.NW);
    }
    const int ebits = std::max(1, hostbig::bit_length(e));
    ewords = (ebits + 31) / 32;
    vmn_garray* r = nullptr;
    VMN_TRY(new_garray(g, n, &r));
    if (n == 0) {
        *out = r;
        return VMN_OK;
    }
    DevTmp ew(ctx), fw(ctx);
    int rc = ew.alloc(ewords * sizeof(uint32_t));
    if (rc == VMN_OK) rc = h2d(ctx, ew.p, e.data(), ewords * sizeof(uint32_t));
    if (rc == VMN_OK) rc = fw.alloc(n * (size_t)g->Q.NW * sizeof(uint32_t));
    if (rc == VMN_OK) rc = to_words(ctx, g->Q, f->d, n, fw.as<uint32_t>());
    if (rc == VMN_OK && g->P.ec) {
        // curves (round 4): one chain of doublings for both scalar multiplications, k_ec_mulvar2
        const vmn_modulus& m = g->P;
        const int wbits = 4;
        const unsigned grid = std::min<unsigned>(grid_for(n), (unsigned)(ctx->num_cus * 2));
        const size_t tab_bytes = (size_t)grid * BLOCK * ((size_t)2 << wbits) * elem_words(m) * sizeof(uint32_t);
        rc = ensure_scratch(ctx, tab_bytes);
        if (rc == VMN_OK) {
            const int nw1 = (ebits + wbits - 1) / wbits, nw2 = (fbits + wbits - 1) / wbits;
            note_work(ctx, m, (double)n * (EC_DBL * std::max(ebits, fbits) + EC_ADD * (nw1 + nw2 + 2 * ((1 << wbits) - 2))));
            rc = VMN_ERR_ARG;
#define X(S_, NW_)                                                                                                              \
    if (m.ec->S == S_)                                                                                                          \
        rc = launch_light(ctx, "modpow", k_ec_mulvar2<S_>, grid, r->d, (const uint32_t*)x->d, (const uint32_t*)ew.as<uint32_t>(), \
                          ewords, ebits, (const uint32_t*)y->d, (const uint32_t*)fw.as<uint32_t>(), g->Q.NW, (size_t)g->Q.NW,     \
                          fbits, wbits, n, ecdev(m.ec), reinterpret_cast<uint32_t*>(ctx->scratch));
            VMN_FOR_CURVES(X)
#undef X
        }
    } else if (rc == VMN_OK) {
        const vmn_modulus& m = geom(ctx, g->P, n);
        const int wbits = std::min(pick_window(std::max(ebits, fbits)), 5);       // two tables per lane
        unsigned max_blocks = (unsigned)(ctx->num_cus * blocks_per_cu(m));
        if (const char* mb = getenv("VMN_MODPOW_MAX_BLOCKS")) {       // (test hook, see modpow_words)
            const int v = atoi(mb);
            if (v >= 1) max_blocks = std::min<unsigned>(max_blocks, (unsigned)v);
        }
        const unsigned grid = std::min<unsigned>(egrid(m, n), max_blocks);
        const size_t tab_bytes = (size_t)grid * (BLOCK / m.LPE) * ((size_t)2 << wbits) * elem_words(m) * sizeof(uint32_t);
        rc = ensure_scratch(ctx, tab_bytes);
        // more than one round of tiles: in phases from a queue of units (k_modpow2_phased; modpow_words has the reasons)
        const size_t epb = (size_t)(BLOCK / m.LPE), ntiles = (n + epb - 1) / epb;
        const int nwin_all = (std::max(ebits, fbits) + wbits - 1) / wbits;
        static const int phases_env2 = [] {
            const char* e = getenv("VMN_MODPOW_PHASES");
            return e && *e ? std::max(1, atoi(e)) : 16;
        }();
        const int phases = ntiles > (size_t)max_blocks && ntiles < ((size_t)1 << 26) ? std::min(phases_env2, std::max(1, nwin_all)) : 1;
        const size_t ptab_bytes = ntiles * epb * ((size_t)2 << wbits) * elem_words(m) * sizeof(uint32_t);
        DevTmp ptab(ctx), sync_words(ctx);
        bool phased = rc == VMN_OK && phases > 1 && ptab_bytes <= ((size_t)64 << 30);
        if (phased && ptab.alloc(ptab_bytes) != VMN_OK) phased = false;
        if (phased) rc = sync_words.alloc((ntiles + 1) * sizeof(uint32_t));
        if (phased && rc == VMN_OK) rc = dev_zero(ctx, sync_words.p, (ntiles + 1) * sizeof(uint32_t));
        if (phased && rc == VMN_OK) {
            const int nw1 = (ebits + wbits - 1) / wbits, nw2 = (fbits + wbits - 1) / wbits;
            note_work(ctx, m, (double)n * (nw1 + nw2 + 2 * ((1 << wbits) - 2)), (double)n * (std::max(nw1, nw2) - 1) * wbits);
            uint32_t* queue = sync_words.as<uint32_t>();
            rc = VMN_ERR_ARG;
#define X(S_, NW_, LPE_)                                                                                                       \
    if (m.S == S_)                                                                                                             \
        rc = launch(ctx, "modpow", k_modpow2_phased<Cfg<S_, LPE_>>, max_blocks, lds_bytes(m), r->d, (const uint32_t*)x->d,        \
                    (const uint32_t*)ew.as<uint32_t>(), ewords, (size_t)0, ebits, (const uint32_t*)y->d,                         \
                    (const uint32_t*)fw.as<uint32_t>(), g->Q.NW, (size_t)g->Q.NW, fbits, wbits, n, m.d_n, m.n0inv, m.d_one,    \
                    ptab.as<uint32_t>(), phases, queue, queue + 1);
            VMN_FOR_SIZES(X)
#undef X
        } else if (rc == VMN_OK) {
            const int nw1 = (ebits + wbits - 1) / wbits, nw2 = (fbits + wbits - 1) / wbits;
            note_work(ctx, m, (double)n * (nw1 + nw2 + 2 * ((1 << wbits) - 2)), (double)n * (std::max(nw1, nw2) - 1) * wbits);
            rc = VMN_ERR_ARG;
#define X(S_, NW_, LPE_)                                                                                                       \
    if (m.S == S_)                                                                                                             \
        rc = launch(ctx, "modpow", k_modpow2<Cfg<S_, LPE_>>, grid, lds_bytes(m), r->d, (const uint32_t*)x->d,                      \
                    (const uint32_t*)ew.as<uint32_t>(), ewords, (size_t)0, ebits, (const uint32_t*)y->d,                         \
                    (const uint32_t*)fw.as<uint32_t>(), g->Q.NW, (size_t)g->Q.NW, fbits, wbits, n, m.d_n, m.n0inv, m.d_one,    \
                    reinterpret_cast<uint32_t*>(ctx->scratch));
            VMN_FOR_SIZES(X)
#undef X
        }
    }
    if (rc != VMN_OK) {
        vmn_garray_free(r);
        return rc;
    }
    *out = r;
    return VMN_OK;
}

// out_x[i] = x[i]^e and out_y[i] = y[i]^f[i] as ONE launch (k_modpow_jobs): the two powers of a verifier's check (B) in its
// separate form, for arrays too small to fill the device alone.  The longer job's blocks come first in the grid, so that
// they are placed on compute units of their own and the shorter job's blocks double up.  Curves, and arrays whose tiles
// would not all be resident at once, take the two ordinary launches.
extern "C" int vmn_garray_exp_pair(const vmn_garray* x, const uint8_t* e_be, size_t ebytes, const vmn_garray* y, const vmn_rarray* f,
                                   int fbits, vmn_garray** out_x, vmn_garray** out_y) {
    ARG_CHECK(x && e_be && ebytes > 0 && y && f && out_x && out_y, "null argument");
    ARG_CHECK(x->grp == y->grp && x->grp == f->grp && y->n == f->n, "arrays differ in group or size");
    vmn_group* g = x->grp;
    vmn_ctx* ctx = LANE(g->ctx);
    {
        VMN_ENTER(ctx);
        if (fbits <= 0 || fbits > g->Q.nbits) fbits = g->Q.nbits;
        const size_t nx = x->n, ny = y->n;
        // the geometry of one job alone at the mean length; eight lanes per element only while the two grids together are
        // resident at once (a second round of tiles costs a whole chain).  Four lanes per element stay ahead of one even in two
        // rounds (profiles/r03_pair_sweep.txt: 18 000 elements 12.5 ms against 14.7 ms).
        const vmn_modulus* mp = &geom(ctx, g->P, (nx + ny + 1) / 2);
        if (mp == g->P.wide8 && g->P.wide && (size_t)egrid(*mp, nx) + egrid(*mp, ny) > (size_t)ctx->num_cus * blocks_per_cu(*mp)) mp = g->P.wide;
        if (mp == g->P.wide && (size_t)egrid(*mp, nx) + egrid(*mp, ny) > 2 * (size_t)ctx->num_cus * blocks_per_cu(*mp)) mp = &g->P;   // (a third round: no)
        const vmn_modulus& m = *mp;
        size_t tiles = (size_t)egrid(m, nx) + egrid(m, ny);
        // Mixed form (2048-bit rows): when four lanes per element were chosen and the LONGER job's tiles at eight lanes still fit
        // beside the shorter job's at four, the longer chain -- which sets the time of the launch -- gets the eight
        // (k_modpow_jobs_mixed; VMN_PAIR_MIXED=0 turns it off).
        static const bool mixed_off = [] {
            const char* e = getenv("VMN_PAIR_MIXED");
            return e && *e == '0';
        }();
        const bool mixed_fits = !mixed_off && !g->P.ec && mp == g->P.wide && g->P.wide8 && g->P.S == 74 && nx > 0 && ny > 0;
        if (!g->P.ec && nx > 0 && ny > 0 && tiles <= (size_t)ctx->num_cus * 2 * blocks_per_cu(m)) {
            int ewords = (int)((ebytes + 3) / 4);
            Big e = hostbig::from_be(e_be, ebytes, ewords);
            const int ebits = std::max(1, hostbig::bit_length(e));
            ewords = (ebits + 31) / 32;
            const int wbits = std::min(pick_window(std::max(ebits, fbits)), 5);
            vmn_garray *rx = nullptr, *ry = nullptr;
            VMN_TRY(new_garray(g, nx, &rx));
            int rc = new_garray(g, ny, &ry);
            DevTmp ew(ctx), fw(ctx);
            if (rc == VMN_OK) rc = ew.alloc(ewords * sizeof(uint32_t));
            if (rc == VMN_OK) rc = h2d(ctx, ew.p, e.data(), ewords * sizeof(uint32_t));
            if (rc == VMN_OK) rc = fw.alloc(ny * (size_t)g->Q.NW * sizeof(uint32_t));
            if (rc == VMN_OK) rc = to_words(ctx, g->Q, f->d, ny, fw.as<uint32_t>());
            if (rc == VMN_OK) rc = ensure_scratch(ctx, tiles * (BLOCK / m.LPE) * ((size_t)1 << wbits) * elem_words(m) * sizeof(uint32_t));
            if (rc == VMN_OK) {
                ModpowJob jx{rx->d, x->d, ew.as<uint32_t>(), ewords, 0, ebits, nx};
                ModpowJob jy{ry->d, y->d, fw.as<uint32_t>(), g->Q.NW, (size_t)g->Q.NW, fbits, ny};
                const bool y_first = (double)fbits * ny >= (double)ebits * nx;
                const ModpowJob& j0 = y_first ? jy : jx;
                const ModpowJob& j1 = y_first ? jx : jy;
                const int nwx = (ebits + wbits - 1) / wbits, nwy = (fbits + wbits - 1) / wbits;
                note_work(ctx, m, (double)nx * (nwx - 1 + (1 << wbits) - 2) + (double)ny * (nwy - 1 + (1 << wbits) - 2),
                          ((double)nx * (nwx - 1) + (double)ny * (nwy - 1)) * wbits);
                const vmn_modulus& m8 = mixed_fits ? *g->P.wide8 : m;
                const size_t tiles_mixed = (size_t)egrid(m8, j0.n) + egrid(m, j1.n);
                // (up to one and a half workgroups per compute unit: beyond, too many eight-lane tiles share their SIMDs with another
                // wave -- 8 000 elements each 5.7 against 6.6 ms, 10 000 8.5 against 6.9, profiles/r03_pair_sweep.txt)
                if (mixed_fits && 2 * tiles_mixed <= 3 * (size_t)ctx->num_cus) {
                    // (scratch: tiles_mixed blocks x the larger tile; sized above for `tiles` blocks of the four-lane tile)
                    rc = ensure_scratch(ctx, tiles_mixed * (BLOCK / m.LPE) * ((size_t)1 << wbits) * elem_words(m) * sizeof(uint32_t));
                    if (rc == VMN_OK)
                        rc = launch(ctx, "modpow", k_modpow_jobs_mixed<Cfg<80, 8>, Cfg<76, 4>>, (unsigned)tiles_mixed,
                                    std::max(lds_bytes(m8), lds_bytes(m)), j0, j1, egrid(m8, j0.n), wbits, m.d_n, m.n0inv, m.d_one,
                                    reinterpret_cast<uint32_t*>(ctx->scratch));
                } else {
                rc = VMN_ERR_ARG;
#define X(S_, NW_, LPE_)                                                                                                     \
    if (m.S == S_)                                                                                                           \
        rc = launch(ctx, "modpow", k_modpow_jobs<Cfg<S_, LPE_>>, (unsigned)tiles, lds_bytes(m), j0, j1, egrid(m, j0.n), wbits, m.d_n, \
                    m.n0inv, m.d_one, reinterpret_cast<uint32_t*>(ctx->scratch));
                VMN_FOR_SIZES(X)
#undef X
                }
            }
            if (rc != VMN_OK) {
                vmn_garray_free(rx);
                if (ry) vmn_garray_free(ry);
                return rc;
            }
            *out_x = rx;
            *out_y = ry;
            return VMN_OK;
        }
    }
    vmn_garray* rx = nullptr;
    VMN_TRY(vmn_garray_exp_scalar(x, e_be, ebytes, &rx));
    int rc = vmn_garray_exp_array(y, f, fbits, out_y);
    if (rc != VMN_OK) {
        vmn_garray_free(rx);
        return rc;
    }
    *out_x = rx;
    return VMN_OK;
}

// ================================================================================================
// second part: K2 fixed base, K3 multi-exponentiation, K5 reductions, K6 compare, K7 movement,
// K8 ring operations
// ================================================================================================
// ---- K6 ------------------------------------------------------------------------------------------
static int compare_arrays(vmn_ctx* ctx, const vmn_modulus& m, const uint32_t* x, const uint32_t* y, size_t n, int* equal) {
    *equal = 1;
    if (n == 0) return VMN_OK;
    VMN_TRY(dev_zero(ctx, ctx->flags, sizeof(uint32_t)));
    if (m.ec) {                        // Jacobian rows: equality of group elements, not of bytes
        int rc = VMN_ERR_ARG;
#define X(S_, NW_) \
    if (m.ec->S == S_) rc = launch_light(ctx, "compare", k_ec_equal<S_>, grid_for(n), x, y, n, ecdev(m.ec), ctx->flags);
        VMN_FOR_CURVES(X)
#undef X
        VMN_TRY(rc);
        uint32_t fl = 0;
        VMN_TRY(read_flag(ctx, &fl));
        *equal = fl ? 0 : 1;
        return VMN_OK;
    }
    size_t nchunks = n * elem_words(m) / 4;
    VMN_TRY(launch_light(ctx, "compare", k_compare, light_grid(ctx, nchunks), reinterpret_cast<const uint4*>(x),
                         reinterpret_cast<const uint4*>(y), nchunks, ctx->flags));
    uint32_t fl = 0;
    VMN_TRY(read_flag(ctx, &fl));
    *equal = fl ? 0 : 1;
    return VMN_OK;
}

extern "C" int vmn_garray_equals(const vmn_garray* x, const vmn_garray* y, int* equal) {
    ARG_CHECK(x && y && equal, "null argument");
    ARG_CHECK(x->grp == y->grp, "arrays differ in group");
    VMN_ENTER(LANE(x->grp->ctx));
    if (x->n != y->n) {
        *equal = 0;
        return VMN_OK;
    }
    return compare_arrays(LANE(x->grp->ctx), x->grp->P, x->d, y->d, x->n, equal);
}
extern "C" int vmn_rarray_equals(const vmn_rarray* x, const vmn_rarray* y, int* equal) {
    ARG_CHECK(x && y && equal, "null argument");
    ARG_CHECK(x->grp == y->grp, "arrays differ in group");
    VMN_ENTER(LANE(x->grp->ctx));
    if (x->n != y->n) {
        *equal = 0;
        return VMN_OK;
    }
    return compare_arrays(LANE(x->grp->ctx), x->grp->Q, x->d, y->d, x->n, equal);
}

// ---- K7 ------------------------------------------------------------------------------------------
// out (n_out rows) = gather of in by host index list; 0xffffffff selects `fill` (device row)
static int gather_rows(vmn_ctx* ctx, const vmn_modulus& m, const uint32_t* in, const std::vector<uint32_t>& idx,
                       const uint32_t* d_fill, uint32_t* out) {
    size_t n_out = idx.size();
    if (n_out == 0) return VMN_OK;
    DevTmp didx(ctx);
    VMN_TRY(didx.alloc(n_out * sizeof(uint32_t)));
    VMN_TRY(h2d(ctx, didx.p, idx.data(), n_out * sizeof(uint32_t)));
    int cpr = (int)(elem_words(m) / 4);
    return launch_light(ctx, "gather", k_gather, light_grid(ctx, n_out * cpr), reinterpret_cast<uint4*>(out),
                        reinterpret_cast<const uint4*>(in), didx.as<uint32_t>(), reinterpret_cast<const uint4*>(d_fill),
                        n_out, cpr);
}

template <typename Arr>
static int arr_gather(const Arr* x, const vmn_modulus& m, const std::vector<uint32_t>& idx, const uint32_t* d_fill,
                      int (*mk)(vmn_group*, size_t, Arr**), void (*fr)(Arr*), Arr** out) {
    Arr* r = nullptr;
    VMN_TRY(mk(x->grp, idx.size(), &r));
    int rc = gather_rows(LANE(x->grp->ctx), m, x->d, idx, d_fill, r->d);
    if (rc != VMN_OK) {
        fr(r);
        return rc;
    }
    *out = r;
    return VMN_OK;
}

extern "C" int vmn_garray_gather(const vmn_garray* x, const uint32_t* idx_host, size_t n_out, vmn_garray** out) {
    ARG_CHECK(x && out && (idx_host || n_out == 0), "null argument");
    VMN_ENTER(LANE(x->grp->ctx));
    std::vector<uint32_t> idx(idx_host, idx_host + n_out);
    for (uint32_t v : idx) ARG_CHECK(v < x->n, "gather index out of range");
    return arr_gather<vmn_garray>(x, x->grp->P, idx, nullptr, new_garray, vmn_garray_free, out);
}
extern "C" int vmn_rarray_gather(const vmn_rarray* x, const uint32_t* idx_host, size_t n_out, vmn_rarray** out) {
    ARG_CHECK(x && out && (idx_host || n_out == 0), "null argument");
    VMN_ENTER(LANE(x->grp->ctx));
    std::vector<uint32_t> idx(idx_host, idx_host + n_out);
    for (uint32_t v : idx) ARG_CHECK(v < x->n, "gather index out of range");
    return arr_gather<vmn_rarray>(x, x->grp->Q, idx, nullptr, new_rarray, vmn_rarray_free, out);
}
extern "C" int vmn_garray_permute(const vmn_garray* x, const uint32_t* perm_host, vmn_garray** out) {
    ARG_CHECK(x, "null argument");
    return vmn_garray_gather(x, perm_host, x->n, out);
}
extern "C" int vmn_rarray_permute(const vmn_rarray* x, const uint32_t* perm_host, vmn_rarray** out) {
    ARG_CHECK(x, "null argument");
    return vmn_rarray_gather(x, perm_host, x->n, out);
}
extern "C" int vmn_garray_shift_push(const vmn_garray* x, const uint8_t* el_be, vmn_garray** out) {
    ARG_CHECK(x && el_be && out, "null argument");
    vmn_group* g = x->grp;
    VMN_ENTER(LANE(g->ctx));
    uint32_t* d_el = nullptr;
    VMN_TRY(import_one(LANE(g->ctx), g->P, g->nbytes, el_be, &d_el));
    std::vector<uint32_t> idx(x->n);
    for (size_t i = 0; i < x->n; ++i) idx[i] = i == 0 ? 0xffffffffu : (uint32_t)(i - 1);
    int rc = arr_gather<vmn_garray>(x, g->P, idx, d_el, new_garray, vmn_garray_free, out);
    free_one(LANE(g->ctx), g->P, d_el);
    return rc;
}
extern "C" int vmn_rarray_shift_push(const vmn_rarray* x, const uint8_t* el_be, vmn_rarray** out) {
    ARG_CHECK(x && el_be && out, "null argument");
    vmn_group* g = x->grp;
    VMN_ENTER(LANE(g->ctx));
    uint32_t* d_el = nullptr;
    VMN_TRY(import_one(LANE(g->ctx), g->Q, g->xbytes, el_be, &d_el));
    std::vector<uint32_t> idx(x->n);
    for (size_t i = 0; i < x->n; ++i) idx[i] = i == 0 ? 0xffffffffu : (uint32_t)(i - 1);
    int rc = arr_gather<vmn_rarray>(x, g->Q, idx, d_el, new_rarray, vmn_rarray_free, out);
    free_one(LANE(g->ctx), g->Q, d_el);
    return rc;
}
extern "C" int vmn_garray_copy_range(const vmn_garray* x, size_t from, size_t to, vmn_garray** out) {
    ARG_CHECK(x && out, "null argument");
    ARG_CHECK(from <= to && to <= x->n, "range out of bounds");
    vmn_group* g = x->grp;
    VMN_ENTER(LANE(g->ctx));
    vmn_garray* r = nullptr;
    VMN_TRY(new_garray(g, to - from, &r));
    if (to > from) {
        hipError_t he = hipMemcpyAsync(r->d, x->d + from * elem_words(g->P), (to - from) * elem_words(g->P) * sizeof(uint32_t),
                                       hipMemcpyDeviceToDevice, LANE(g->ctx)->stream);
        if (he != hipSuccess) {
            vmn_garray_free(r);
            set_error("copy failed: %s", hipGetErrorString(he));
            return VMN_ERR_DEVICE;
        }
    }
    *out = r;
    return VMN_OK;
}
extern "C" int vmn_garray_extract(const vmn_garray* x, const uint8_t* keep_host, vmn_garray** out) {
    ARG_CHECK(x && out && (keep_host || x->n == 0), "null argument");
    VMN_ENTER(LANE(x->grp->ctx));
    std::vector<uint32_t> idx;
    for (size_t i = 0; i < x->n; ++i) {
        if (keep_host[i]) idx.push_back((uint32_t)i);
    }
    return arr_gather<vmn_garray>(x, x->grp->P, idx, nullptr, new_garray, vmn_garray_free, out);
}
extern "C" int vmn_garray_get(const vmn_garray* x, size_t i, uint8_t* out_be) {
    ARG_CHECK(x && out_be, "null argument");
    ARG_CHECK(i < x->n, "index out of range");
    VMN_ENTER(LANE(x->grp->ctx));
    return export_be(LANE(x->grp->ctx), x->grp->P, x->grp->nbytes, x->d + i * elem_words(x->grp->P), 1, out_be);
}

extern "C" int vmn_rarray_get(const vmn_rarray* x, size_t i, uint8_t* out_be) {
    ARG_CHECK(x && out_be, "null argument");
    ARG_CHECK(i < x->n, "index out of range");
    VMN_ENTER(LANE(x->grp->ctx));
    return export_be(LANE(x->grp->ctx), x->grp->Q, x->grp->xbytes, x->d + i * elem_words(x->grp->Q), 1, out_be);
}
extern "C" int vmn_rarray_copy_range(const vmn_rarray* x, size_t from, size_t to, vmn_rarray** out) {
    ARG_CHECK(x && out, "null argument");
    ARG_CHECK(from <= to && to <= x->n, "range out of bounds");
    vmn_group* g = x->grp;
    VMN_ENTER(LANE(g->ctx));
    vmn_rarray* r = nullptr;
    VMN_TRY(new_rarray(g, to - from, &r));
    if (to > from) {
        hipError_t he = hipMemcpyAsync(r->d, x->d + from * elem_words(g->Q), (to - from) * elem_words(g->Q) * sizeof(uint32_t),
                                       hipMemcpyDeviceToDevice, LANE(g->ctx)->stream);
        if (he != hipSuccess) {
            vmn_rarray_free(r);
            set_error("copy failed: %s", hipGetErrorString(he));
            return VMN_ERR_DEVICE;
        }
    }
    *out = r;
    return VMN_OK;
}

// ---- K5 / reductions -----------------------------------------------------------------------------
// Reduce nseg segments of len elements each to nseg single elements (d_out: nseg rows).
// mul = product, else sum.  Work buffers ping-pong inside one temporary.
static int reduce_segments(vmn_ctx* ctx, const vmn_modulus& m, const uint32_t* x, size_t len, size_t nseg, bool mul,
                           uint32_t* d_out) {
    const size_t Wd = elem_words(m);
    if (len == 0) {       // empty product = one, empty sum = zero
        std::vector<uint32_t> row(Wd, 0);
        if (mul) {
            VMN_TRY(d2h(ctx, row.data(), m.d_one, Wd * sizeof(uint32_t)));
        }
        for (size_t s = 0; s < nseg; ++s) VMN_TRY(h2d(ctx, d_out + s * Wd, row.data(), Wd * sizeof(uint32_t)));
        return VMN_OK;
    }
    const size_t max_lanes = (size_t)ctx->num_cus * blocks_per_cu(m) * (BLOCK / m.LPE);
    // first pass: as many lanes as the machine holds; then every pass folds R values per lane.  The tail passes are short
    // launches whose cost is the launch itself, so R trades passes (17 when halving from 131 072 lanes) against the chain of
    // R - 1 operations inside a lane: 16 for the cheap ones (sums and products mod q), 4 for group elements.
    const size_t R = (!m.ec && m.S <= 20) ? 16 : 4;
    size_t L = std::min(std::max<size_t>(max_lanes / std::max<size_t>(nseg, 1), 1), (len + 1) / 2);
    if (len == 1) L = 1;
    DevTmp buf(ctx);
    size_t cap = nseg * std::max<size_t>(L, 1);
    VMN_TRY(buf.alloc(2 * cap * Wd * sizeof(uint32_t)));
    uint32_t* ping = buf.as<uint32_t>();
    uint32_t* pong = ping + cap * Wd;
    const uint32_t* src = x;
    size_t cur = len;
    while (true) {
        uint32_t* dst = (L == 1) ? d_out : ping;
        if (mul || m.ec) note_work(ctx, m, (m.ec ? EC_ADD : 1.0) * (double)nseg * (double)(cur - L));
        int rc = VMN_ERR_ARG;
        if (m.ec) {
#define X(S_, NW_) \
    if (m.ec->S == S_) rc = launch_light(ctx, "reduce", k_ec_reduce<S_>, grid_for(nseg * L), dst, src, cur, L, nseg, ecdev(m.ec));
            VMN_FOR_CURVES(X)
#undef X
        } else {
#define X(S_, NW_, LPE_)                                                                                                      \
    if (m.S == S_) {                                                                                                    \
        rc = mul ? launch(ctx, "reduce", k_reduce_strided<Cfg<S_, LPE_>, true>, egrid(m, nseg * L), lds_bytes(m), dst, src, cur, L, \
                          nseg, m.d_n, m.n0inv)                                                                         \
                 : launch(ctx, "reduce", k_reduce_strided<Cfg<S_, LPE_>, false>, egrid(m, nseg * L), lds_bytes(m), dst, src, cur, \
                          L, nseg, m.d_n, m.n0inv);                                                                     \
    }
        VMN_DISPATCH((nseg * L) / 8, X)        /* the tail levels of a reduction are latency-bound: wide up to 8 x the threshold */
#undef X
        }
        VMN_TRY(rc);
        if (L == 1) break;
        src = dst;
        cur = L;
        L = (cur + R - 1) / R;
        std::swap(ping, pong);
    }
    return VMN_OK;
}

static int reduce_to_host(vmn_ctx* ctx, const vmn_modulus& m, size_t nbytes, const uint32_t* x, size_t n, bool mul,
                          uint8_t* out_be) {
    DevTmp one(ctx);
    VMN_TRY(one.alloc(elem_words(m) * sizeof(uint32_t)));
    VMN_TRY(reduce_segments(ctx, m, x, n, 1, mul, one.as<uint32_t>()));
    return export_be(ctx, m, nbytes, one.as<uint32_t>(), 1, out_be);
}

extern "C" int vmn_garray_prod(const vmn_garray* x, uint8_t* out_be) {
    ARG_CHECK(x && out_be, "null argument");
    VMN_ENTER(LANE(x->grp->ctx));
    return reduce_to_host(LANE(x->grp->ctx), x->grp->P, x->grp->nbytes, x->d, x->n, true, out_be);
}
extern "C" int vmn_rarray_prod(const vmn_rarray* x, uint8_t* out_be) {
    ARG_CHECK(x && out_be, "null argument");
    VMN_ENTER(LANE(x->grp->ctx));
    return reduce_to_host(LANE(x->grp->ctx), x->grp->Q, x->grp->xbytes, x->d, x->n, true, out_be);
}
extern "C" int vmn_rarray_sum(const vmn_rarray* x, uint8_t* out_be) {
    ARG_CHECK(x && out_be, "null argument");
    VMN_ENTER(LANE(x->grp->ctx));
    return reduce_to_host(LANE(x->grp->ctx), x->grp->Q, x->grp->xbytes, x->d, x->n, false, out_be);
}
extern "C" int vmn_rarray_inner_product(const vmn_rarray* x, const vmn_rarray* y, uint8_t* out_be) {
    ARG_CHECK(x && y && out_be, "null argument");
    ARG_CHECK(x->grp == y->grp && x->n == y->n, "arrays differ in group or size");
    vmn_group* g = x->grp;
    VMN_ENTER(LANE(g->ctx));
    DevTmp prod(LANE(g->ctx));
    VMN_TRY(prod.alloc(std::max<size_t>(x->n, 1) * elem_words(g->Q) * sizeof(uint32_t)));
    VMN_TRY(mul_arrays(LANE(g->ctx), g->Q, x->d, y->d, elem_words(g->Q), x->n, prod.as<uint32_t>()));
    return reduce_to_host(LANE(g->ctx), g->Q, g->xbytes, prod.as<uint32_t>(), x->n, false, out_be);
}

// k scalars in one round trip: out[i] = <xs[i], ys[i]> mod q, or the sum of xs[i] where ys[i] is null -- a prover's reply takes
// <r, e'>, sum r and <s_c, e> per column (PoSBasicTW.java:856-888); each alone ends in a read-back that drains the stream.
extern "C" int vmn_rarray_inner_products(const vmn_rarray* const* xs, const vmn_rarray* const* ys, size_t k, uint8_t* out_be) {
    ARG_CHECK(xs && ys && k > 0 && out_be, "null argument");
    ARG_CHECK(xs[0], "null array");
    vmn_group* g = xs[0]->grp;
    for (size_t i = 0; i < k; ++i)
        ARG_CHECK(xs[i] && xs[i]->grp == g && (!ys[i] || (ys[i]->grp == g && ys[i]->n == xs[i]->n)), "arrays differ in group or size");
    vmn_ctx* ctx = LANE(g->ctx);
    VMN_ENTER(ctx);
    const size_t Wd = elem_words(g->Q);
    DevTmp res(ctx), prod(ctx);
    VMN_TRY(res.alloc(k * Wd * sizeof(uint32_t)));
    for (size_t i = 0; i < k; ++i) {
        const uint32_t* src = xs[i]->d;
        if (ys[i]) {
            VMN_TRY(prod.alloc(std::max<size_t>(xs[i]->n, 1) * Wd * sizeof(uint32_t)));
            VMN_TRY(mul_arrays(ctx, g->Q, xs[i]->d, ys[i]->d, Wd, xs[i]->n, prod.as<uint32_t>()));
            src = prod.as<uint32_t>();
        }
        VMN_TRY(reduce_segments(ctx, g->Q, src, xs[i]->n, 1, false, res.as<uint32_t>() + i * Wd));
    }
    return export_be(ctx, g->Q, g->xbytes, res.as<uint32_t>(), k, out_be);
}

// ---- K8 element-wise -------------------------------------------------------------------------------
static int ring_elementwise(vmn_ctx* ctx, const vmn_modulus& m, const uint32_t* x, const uint32_t* y, const uint32_t* v,
                            int op, size_t n, uint32_t* out) {
    if (n == 0) return VMN_OK;
    if (op >= 2) note_work(ctx, m, (double)n);
    int rc = VMN_ERR_ARG;
#define X(S_, NW_, LPE_) \
    if (m.S == S_) rc = launch(ctx, "ring", k_ring_elementwise<Cfg<S_, LPE_>>, egrid(m, n), lds_bytes(m), out, x, y, v, op, n, m.d_n, m.n0inv);
    VMN_DISPATCH(n, X)
#undef X
    return rc;
}

extern "C" int vmn_rarray_mul(const vmn_rarray* x, const vmn_rarray* y, vmn_rarray** out) {
    ARG_CHECK(x && y && out, "null argument");
    ARG_CHECK(x->grp == y->grp && x->n == y->n, "arrays differ in group or size");
    vmn_group* g = x->grp;
    VMN_ENTER(LANE(g->ctx));
    vmn_rarray* r = nullptr;
    VMN_TRY(new_rarray(g, x->n, &r));
    int rc = mul_arrays(LANE(g->ctx), g->Q, x->d, y->d, elem_words(g->Q), x->n, r->d);
    if (rc != VMN_OK) {
        vmn_rarray_free(r);
        return rc;
    }
    *out = r;
    return VMN_OK;
}
extern "C" int vmn_rarray_add(const vmn_rarray* x, const vmn_rarray* y, vmn_rarray** out) {
    ARG_CHECK(x && y && out, "null argument");
    ARG_CHECK(x->grp == y->grp && x->n == y->n, "arrays differ in group or size");
    vmn_group* g = x->grp;
    VMN_ENTER(LANE(g->ctx));
    vmn_rarray* r = nullptr;
    VMN_TRY(new_rarray(g, x->n, &r));
    int rc = ring_elementwise(LANE(g->ctx), g->Q, x->d, y->d, nullptr, 0, x->n, r->d);
    if (rc != VMN_OK) {
        vmn_rarray_free(r);
        return rc;
    }
    *out = r;
    return VMN_OK;
}
extern "C" int vmn_rarray_neg(const vmn_rarray* x, vmn_rarray** out) {
    ARG_CHECK(x && out, "null argument");
    vmn_group* g = x->grp;
    VMN_ENTER(LANE(g->ctx));
    vmn_rarray* r = nullptr;
    VMN_TRY(new_rarray(g, x->n, &r));
    int rc = ring_elementwise(LANE(g->ctx), g->Q, x->d, nullptr, nullptr, 1, x->n, r->d);
    if (rc != VMN_OK) {
        vmn_rarray_free(r);
        return rc;
    }
    *out = r;
    return VMN_OK;
}
extern "C" int vmn_rarray_mul_add(const vmn_rarray* x, const uint8_t* v_be, const vmn_rarray* y, vmn_rarray** out) {
    ARG_CHECK(x && v_be && out, "null argument");
    ARG_CHECK(!y || (x->grp == y->grp && x->n == y->n), "arrays differ in group or size");
    vmn_group* g = x->grp;
    VMN_ENTER(LANE(g->ctx));
    uint32_t* d_v = nullptr;
    VMN_TRY(import_one(LANE(g->ctx), g->Q, g->xbytes, v_be, &d_v));
    vmn_rarray* r = nullptr;
    int rc = new_rarray(g, x->n, &r);
    if (rc == VMN_OK) rc = ring_elementwise(LANE(g->ctx), g->Q, x->d, y ? y->d : nullptr, d_v, y ? 2 : 3, x->n, r->d);
    free_one(LANE(g->ctx), g->Q, d_v);
    if (rc != VMN_OK) {
        if (r) vmn_rarray_free(r);
        return rc;
    }
    *out = r;
    return VMN_OK;
}

// ---- K8 scans --------------------------------------------------------------------------------------
// out[i] = out[i-1]*e[i] + b[i]  (b == nullptr: out[i] = out[i-1]*e[i], starting from one), per segment.
// Chunked three-phase scan; the totals of one level are the inputs of the next (same recurrence).
// Chunk length of the three-phase scans: a lane walks `chunk` elements sequentially, so short chunks give more
// lanes (latency hiding) and long chunks fewer levels (work: 2n(1 + 1/chunk + ...)).  `lanes_wanted` is what fills
// the chip for the kernel family; the chunk shrinks (down to 4) until the top level has that many lanes.
static size_t scan_chunk(size_t n, size_t lanes_wanted) {
    if (const char* env = getenv("VMN_SCAN_CHUNK")) {            // measurement knob
        int c = atoi(env);
        if (c >= 2 && c <= 64) return (size_t)c;
    }
    size_t c = 16;
    while (c > 4 && n / c < lanes_wanted) c >>= 1;
    return c;
}

static int scan_affine(vmn_ctx* ctx, const vmn_modulus& m0, const uint32_t* e, const uint32_t* b, size_t n, size_t seglen,
                       int rev, uint32_t* out) {
    if (n == 0) return VMN_OK;
    const vmn_modulus& m = geom(ctx, m0, n, true);     // (the chunk length below depends on the geometry's lanes per element)
    const size_t Wd = elem_words(m);
    if (seglen == 0 || seglen > n) seglen = n;
    if (m.ec) {                                        // running sums of curve points (b must be null)
        if (b) return VMN_ERR_ARG;
        size_t Cc = scan_chunk(n, (size_t)ctx->num_cus * 4 * 64 * 4);          // light kernels: 4 waves per SIMD
        if (seglen != n) {
            while (Cc > 1 && seglen % Cc) Cc >>= 1;
        }
        if (seglen <= Cc) Cc = seglen;
        size_t nchunks = (n + Cc - 1) / Cc;
        int rc = VMN_ERR_ARG;
        if (seglen <= Cc) {
#define X(S_, NW_)                                                                                                 \
    if (m.ec->S == S_)                                                                                             \
        rc = note_work(ctx, m, EC_ADD * (double)n) ? 0 : launch_light(ctx, "scan", k_ec_scan_apply<S_>, grid_for(nchunks), out, e, (const uint32_t*)nullptr, n, \
                          Cc, seglen, rev, ecdev(m.ec));
            VMN_FOR_CURVES(X)
#undef X
            return rc;
        }
        DevTmp tot(ctx);
        VMN_TRY(tot.alloc(2 * nchunks * Wd * sizeof(uint32_t)));
        uint32_t* Etot = tot.as<uint32_t>();
        uint32_t* inc = Etot + nchunks * Wd;
#define X(S_, NW_) \
    if (m.ec->S == S_) rc = note_work(ctx, m, EC_ADD * (double)n) ? 0 : launch_light(ctx, "scan", k_ec_scan_totals<S_>, grid_for(nchunks), Etot, e, n, Cc, seglen, rev, ecdev(m.ec));
        VMN_FOR_CURVES(X)
#undef X
        VMN_TRY(rc);
        size_t seg_chunks = seglen == n ? nchunks : seglen / Cc;
        VMN_TRY(scan_affine(ctx, m, Etot, nullptr, nchunks, seg_chunks, 0, inc));
        rc = VMN_ERR_ARG;
#define X(S_, NW_)                                                                                                       \
    if (m.ec->S == S_)                                                                                                   \
        rc = note_work(ctx, m, EC_ADD * (double)n) ? 0 : launch_light(ctx, "scan", k_ec_scan_apply<S_>, grid_for(nchunks), out, e, (const uint32_t*)inc, n, Cc, seglen, \
                          rev, ecdev(m.ec));
        VMN_FOR_CURVES(X)
#undef X
        return rc;
    }
    // chunk length: divides seglen when there are several segments
    size_t C = scan_chunk(n, (size_t)ctx->num_cus * blocks_per_cu(m) * (BLOCK / m.LPE));   // one tile per resident workgroup
    if (seglen != n) {
        while (C > 1 && seglen % C) C >>= 1;
    }
    if (seglen <= C) {
        // every segment fits one chunk: a single apply pass with fresh starts
        size_t Cs = seglen;
        int rc = VMN_ERR_ARG;
        size_t nchunks = (n + Cs - 1) / Cs;
#define X(S_, NW_, LPE_)                                                                                                        \
    if (m.S == S_)                                                                                                        \
        rc = note_work(ctx, m, (double)n) ? 0 : launch(ctx, "scan", k_scan_apply<Cfg<S_, LPE_>>, egrid(m, nchunks), lds_bytes(m), out, e, b, (const uint32_t*)nullptr, \
                    n, Cs, seglen, rev, m.d_n, m.n0inv, m.d_one);
        VMN_DISPATCH(nchunks, X)
#undef X
        return rc;
    }
    size_t nchunks = (n + C - 1) / C;
    DevTmp tot(ctx);
    VMN_TRY(tot.alloc(3 * nchunks * Wd * sizeof(uint32_t)));
    uint32_t* Etot = tot.as<uint32_t>();
    uint32_t* Xtot = Etot + nchunks * Wd;
    uint32_t* inc = Xtot + nchunks * Wd;
    int rc = VMN_ERR_ARG;
#define X(S_, NW_, LPE_)                                                                                                      \
    if (m.S == S_) {                                                                                                    \
        rc = note_work(ctx, m, (double)n) ? 0 : launch(ctx, "scan", k_scan_totals<Cfg<S_, LPE_>, false>, egrid(m, nchunks), lds_bytes(m), Etot, e, b, n, C, seglen,  \
                    rev, m.d_n, m.n0inv, m.d_one);                                                                      \
        if (rc == VMN_OK && b)                                                                                          \
            rc = note_work(ctx, m, (double)n) ? 0 : launch(ctx, "scan", k_scan_totals<Cfg<S_, LPE_>, true>, egrid(m, nchunks), lds_bytes(m), Xtot, e, b, n, C,       \
                        seglen, rev, m.d_n, m.n0inv, m.d_one);                                                          \
    }
    VMN_DISPATCH(nchunks, X)
#undef X
    VMN_TRY(rc);
    // inclusive scan over the chunk totals with the same recurrence (segments shrink by C)
    size_t seg_chunks = seglen == n ? nchunks : seglen / C;
    VMN_TRY(scan_affine(ctx, m0, Etot, b ? Xtot : nullptr, nchunks, seg_chunks, 0, inc));
    rc = VMN_ERR_ARG;
#define X(S_, NW_, LPE_)                                                                                                  \
    if (m.S == S_)                                                                                                  \
        rc = note_work(ctx, m, (double)n) ? 0 : launch(ctx, "scan", k_scan_apply<Cfg<S_, LPE_>>, egrid(m, nchunks), lds_bytes(m), out, e, b, (const uint32_t*)inc, n, \
                    C, seglen, rev, m.d_n, m.n0inv, m.d_one);
    VMN_DISPATCH(nchunks, X)
#undef X
    return rc;
}

extern "C" int vmn_rarray_max_bits(const vmn_rarray* x, int* bits) {
    ARG_CHECK(x && bits, "null argument");
    vmn_group* g = x->grp;
    vmn_ctx* ctx = LANE(g->ctx);
    VMN_ENTER(ctx);
    *bits = 0;
    if (x->n == 0) return VMN_OK;
    DevTmp ew(ctx);
    VMN_TRY(ew.alloc(x->n * (size_t)g->Q.NW * sizeof(uint32_t)));
    VMN_TRY(to_words(ctx, g->Q, x->d, x->n, ew.as<uint32_t>()));
    VMN_TRY(dev_zero(ctx, ctx->flags, sizeof(uint32_t)));
    VMN_TRY(launch_light(ctx, "ring", k_words_maxbits, grid_for(x->n), (const uint32_t*)ew.as<uint32_t>(), x->n, g->Q.NW, ctx->flags));
    uint32_t v = 0;
    VMN_TRY(read_flag(ctx, &v));
    *bits = (int)v;
    return VMN_OK;
}

extern "C" int vmn_rarray_rec_lin(const vmn_rarray* b, const vmn_rarray* e, vmn_rarray** out_x, uint8_t* last_be) {
    ARG_CHECK(b && e && out_x, "null argument");
    ARG_CHECK(b->grp == e->grp && b->n == e->n, "arrays differ in group or size");
    vmn_group* g = b->grp;
    VMN_ENTER(LANE(g->ctx));
    vmn_rarray* r = nullptr;
    VMN_TRY(new_rarray(g, b->n, &r));
    int rc = scan_affine(LANE(g->ctx), g->Q, e->d, b->d, b->n, b->n, 0, r->d);
    if (rc == VMN_OK && last_be) {
        if (b->n) rc = export_be(LANE(g->ctx), g->Q, g->xbytes, r->d + (b->n - 1) * elem_words(g->Q), 1, last_be);
        else memset(last_be, 0, g->xbytes);
    }
    if (rc != VMN_OK) {
        vmn_rarray_free(r);
        return rc;
    }
    *out_x = r;
    return VMN_OK;
}
extern "C" int vmn_rarray_prods(const vmn_rarray* e, vmn_rarray** out) {
    ARG_CHECK(e && out, "null argument");
    vmn_group* g = e->grp;
    VMN_ENTER(LANE(g->ctx));
    vmn_rarray* r = nullptr;
    VMN_TRY(new_rarray(g, e->n, &r));
    int rc = scan_affine(LANE(g->ctx), g->Q, e->d, nullptr, e->n, e->n, 0, r->d);
    if (rc != VMN_OK) {
        vmn_rarray_free(r);
        return rc;
    }
    *out = r;
    return VMN_OK;
}

// ---- batch inversion (serves K3' with negative coefficients) ----------------------------------------
extern "C" int vmn_garray_inv(const vmn_garray* x, vmn_garray** out) {
    ARG_CHECK(x && out, "null argument");
    vmn_group* g = x->grp;
    vmn_ctx* ctx = LANE(g->ctx);
    const vmn_modulus& m = g->P;
    const size_t Wd = elem_words(m);
    const size_t n = x->n;
    VMN_ENTER(ctx);
    vmn_garray* r = nullptr;
    VMN_TRY(new_garray(g, n, &r));
    if (n == 0) {
        *out = r;
        return VMN_OK;
    }
    if (m.ec) {
        int rce = VMN_ERR_ARG;
#define X(S_, NW_) \
    if (m.ec->S == S_) rce = launch_light(ctx, "modmul", k_ec_neg<S_>, grid_for(n), r->d, (const uint32_t*)x->d, n, ecdev(m.ec));
        VMN_FOR_CURVES(X)
#undef X
        if (rce != VMN_OK) {
            vmn_garray_free(r);
            return rce;
        }
        *out = r;
        return VMN_OK;
    }
    int rc = VMN_OK;
    {
        // P[i] = x0..xi, S[i] = xi..x(n-1);  inv(x_i) = P[i-1] * S[i+1] * (P[n-1])^-1
        DevTmp ps(ctx), sh(ctx);
        rc = ps.alloc(2 * n * Wd * sizeof(uint32_t));
        if (rc == VMN_OK) rc = sh.alloc(2 * n * Wd * sizeof(uint32_t));
        uint32_t* Pf = ps.as<uint32_t>();
        uint32_t* Sf = Pf + n * Wd;
        uint32_t* Psh = sh.as<uint32_t>();
        uint32_t* Ssh = Psh + n * Wd;
        if (rc == VMN_OK) rc = scan_affine(ctx, m, x->d, nullptr, n, n, 0, Pf);
        if (rc == VMN_OK) rc = scan_affine(ctx, m, x->d, nullptr, n, n, 1, Sf);
        // total -> host, invert with the host Montgomery context (x^(p-2)), back to the device
        std::vector<uint8_t> tbe(g->nbytes);
        if (rc == VMN_OK) rc = export_be(ctx, m, g->nbytes, Pf + (n - 1) * Wd, 1, tbe.data());
        if (rc == VMN_OK) {
            const hostbig::Mont& hm = *m.hm;
            Big t = hostbig::from_be(tbe.data(), g->nbytes, m.NW);
            if (hostbig::is_zero(t)) {
                set_error("vmn_garray_inv: an element is not invertible");
                rc = VMN_ERR_FORMAT;
            } else {
                (void)hm;
                const num64::Mod& h64 = *m.hm64;              // binary Euclid (0.15 ms at 2048 bits; a Fermat power takes 16)
                num64::to_be(h64.inv(num64::from_be(tbe.data(), g->nbytes, h64.nl)), tbe.data(), g->nbytes);
                uint32_t* d_t = nullptr;
                rc = import_one(ctx, m, g->nbytes, tbe.data(), &d_t);
                if (rc == VMN_OK) {
                    std::vector<uint32_t> idx(n);
                    for (size_t i = 0; i < n; ++i) idx[i] = i == 0 ? 0xffffffffu : (uint32_t)(i - 1);
                    if (rc == VMN_OK) rc = gather_rows(ctx, m, Pf, idx, m.d_one, Psh);
                    for (size_t i = 0; i < n; ++i) idx[i] = i + 1 == n ? 0xffffffffu : (uint32_t)(i + 1);
                    if (rc == VMN_OK) rc = gather_rows(ctx, m, Sf, idx, m.d_one, Ssh);
                    if (rc == VMN_OK) rc = mul_arrays(ctx, m, Psh, Ssh, Wd, n, Pf);          // reuse Pf as scratch
                    if (rc == VMN_OK) rc = mul_arrays(ctx, m, Pf, d_t, 0, n, r->d);
                    free_one(ctx, m, d_t);
                }
            }
        }
    }
    if (rc != VMN_OK) {
        vmn_garray_free(r);
        return rc;
    }
    *out = r;
    return VMN_OK;
}

// ================================================================================================
// N1: PRGHeuristic(SHA-256) on the device, the random vector of a proof and independent generators
// ================================================================================================
// A PRG seed: PRGHeuristic over SHA-256 / SHA-384 / SHA-512 takes a seed of the digest's length (minNoSeedBytes), so
// the seed length says which hash it is (elgamal/ProtocolElGamal.java:352-371).  Big-endian 32-bit words.
struct PrgSeed {
    int hash = 256;               // 256, 384, 512
    uint32_t w[16] = {0};
    int digest_bytes() const { return hash / 8; }
};
// digest bytes of block `blk` of PRG(seed): H(seed || uint32_be(blk))
template <int HASH>
__host__ __device__ inline void prg_block_bytes(uint8_t (&dg)[HASH / 8], const uint32_t* seed_words, uint32_t blk) {
    if constexpr (HASH == 256) {
        uint32_t sw[8], d[8];
        for (int k = 0; k < 8; ++k) sw[k] = seed_words[k];
        sha256::prg_block(d, sw, blk);
        for (int b = 0; b < 32; ++b) dg[b] = (uint8_t)(d[b >> 2] >> (24 - 8 * (b & 3)));
    } else {
        uint64_t d[8];
        sha512::prg_block<HASH>(d, seed_words, blk);
        for (int b = 0; b < HASH / 8; ++b) dg[b] = (uint8_t)(d[b >> 3] >> (56 - 8 * (b & 7)));
    }
}
// Row i = bytes [part_off, part_off + part_len) of value number vi of the PRG stream (value vi = stream bytes
// [vi*val_bytes, (vi+1)*val_bytes), first byte masked with top_mask), right-aligned in row_bytes bytes; vi = first + i,
// or idx[i] when an index table is given -- the stream is counter mode, so a shard of the positions (or the rows a
// permutation selects) is generated without the rest (the sharded proofs: DESIGN.md §7).
// One lane per row; a lane hashes the digest-sized blocks its value overlaps (<= val_bytes / digest + 2 compressions).
template <int HASH>
__global__ void __launch_bounds__(256) k_prg_rows(uint8_t* __restrict__ out, size_t row_bytes, size_t val_bytes, uint32_t top_mask,
                                                  size_t part_off, size_t part_len, const uint32_t* __restrict__ seed_words,
                                                  size_t n, size_t first, const uint32_t* __restrict__ idx) {
    constexpr int DB = HASH / 8;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t vi = idx ? (size_t)idx[i] : first + i;
    uint32_t seed[HASH / 32];
#pragma unroll
    for (int k = 0; k < HASH / 32; ++k) seed[k] = seed_words[k];
    uint8_t* row = out + i * row_bytes;
    const size_t pad = row_bytes - part_len;
    for (size_t k = 0; k < pad; ++k) row[k] = 0;
    const size_t v0 = vi * val_bytes;                      // stream offset of the value
    const size_t lo = v0 + part_off, hi = lo + part_len;   // stream range wanted
    for (size_t blk = lo / DB; blk * DB < hi; ++blk) {
        uint8_t dg[DB];
        prg_block_bytes<HASH>(dg, seed, (uint32_t)blk);
        for (int b = 0; b < DB; ++b) {
            size_t pos = blk * DB + b;
            if (pos < lo || pos >= hi) continue;
            uint8_t byte = dg[b];
            if (pos == v0) byte &= (uint8_t)top_mask;
            row[pad + (pos - lo)] = byte;
        }
    }
}

// ECqPGroup.randomElementArray(n, prg, rbitlen), candidate by candidate (the reference derives its independent generators
// with it, P/distr/IndependentGeneratorsRO.java:117-130; the procedure itself is VCR's and is restated from the published
// verifier specification [NOT-IN-REF]): candidate j = the j-th ceil((bits(p) + rbitlen) / 8) bytes of the PRG stream with
// the leading bits cleared, reduced mod p, taken as an x coordinate; it is KEPT when x^3 - 3x + b is a square, and then
// the point is (x, y) with y the SMALLER of the two roots; the i-th element of the array is the i-th kept candidate.
// One lane per candidate: x from the PRG, z = (x^3 - 3x + b)^((p+1)/4) (p = 3 mod 4), kept iff z^2 is that value; the
// caller compacts the kept rows in order.  c_m = 2^(8 cb) mod p in Montgomery form (the value is hi * 2^(8 cb) + lo).
template <int S, int NW, int HASH>
__global__ void __launch_bounds__(BLOCK, ECfg<S>::MINW)
k_ec_random_points(u32* __restrict__ rows, u32* __restrict__ keep, const uint32_t* __restrict__ seed_words, size_t first, size_t m,
                   size_t vb, uint32_t top_mask, size_t cb, const u32* __restrict__ c_m, ECDev E) {
    constexpr int DB = HASH / 8;
    constexpr int ROW = ECfg<S>::ROW;
    using C1 = Cfg<S, 1>;
    size_t j = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j >= m) return;
    uint32_t seed[HASH / 32];
#pragma unroll
    for (int k = 0; k < HASH / 32; ++k) seed[k] = seed_words[k];
    // the candidate's bytes: hi = the vb - cb leading ones (right-aligned in 4 NW bytes), lo = the last cb
    uint8_t hi_b[4 * NW], lo_b[4 * NW];
    for (int k = 0; k < 4 * NW; ++k) hi_b[k] = lo_b[k] = 0;
    const size_t v0 = (first + j) * vb, hb = vb - cb;
    for (size_t blk = v0 / DB; blk * DB < v0 + vb; ++blk) {
        uint8_t dg[DB];
        prg_block_bytes<HASH>(dg, seed, (uint32_t)blk);
        for (int b = 0; b < DB; ++b) {
            size_t pos = blk * DB + b;
            if (pos < v0 || pos >= v0 + vb) continue;
            uint8_t byte = dg[b];
            if (pos == v0) byte &= (uint8_t)top_mask;
            size_t k = pos - v0;                       // index inside the value, big-endian
            if (k < hb) hi_b[4 * NW - hb + k] = byte;
            else lo_b[4 * NW - cb + (k - hb)] = byte;
        }
    }
    Lane<C1> ln(nullptr);
    u32 hi[S], lo[S], rr[S], cm[S], x[S], t[S], t2[S], rhs[S], z[S], bb[S];
    limbs_from_be<C1, NW>(hi, hi_b, 4L * NW, ln);
    limbs_from_be<C1, NW>(lo, lo_b, 4L * NW, ln);
#pragma unroll
    for (int k = 0; k < S; ++k) {
        rr[k] = E.rr[k];
        cm[k] = c_m[k];
        bb[k] = E.b[k];
    }
    f_mul<S>(lo, lo, rr, E);                           // Montgomery form (any value below R is reduced on the way)
    f_mul<S>(hi, hi, rr, E);
    f_mul<S>(t, hi, cm, E);
    f_add<S>(x, t, lo);                                // x = hi * 2^(8 cb) + lo  mod p
    f_canon<S>(x, x, E);
    f_sqr<S>(t, x, E);
    f_mul<S>(t2, t, x, E);                             // x^3
    f_small<S, 3>(t, x);
    f_add<S>(rhs, t2, bb);
    f_sub<S>(rhs, rhs, t, E);                          // x^3 - 3x + b
    f_canon<S>(rhs, rhs, E);
    f_sqrt<S>(z, rhs, E);
    f_sqr<S>(t, z, E);
    f_sub<S, true>(t2, t, rhs, E);
    const bool ok = f_is_zero<S>(t2, E);
    // the smaller root: compare the standard representatives of z and p - z
    u32 one1[S], zs[S], zn[S], pp[S];
#pragma unroll
    for (int k = 0; k < S; ++k) {
        one1[k] = k == 0 ? 1u : 0u;
        pp[k] = E.p[k];
    }
    f_mul<S>(zs, z, one1, E);                          // z / R: standard representative, < 2p
    {
        u32 d[S];
        if (borrow_sweep<S>(d, zs, pp, 0) == 0) {
#pragma unroll
            for (int k = 0; k < S; ++k) zs[k] = d[k];
        }
    }
    borrow_sweep<S>(zn, pp, zs, 0);                    // p - z  (z != 0 unless rhs = 0: then both are 0 mod p)
    bool neg_smaller = false;
    for (int k = S - 1; k >= 0; --k) {
        if (zn[k] != zs[k]) {
            neg_smaller = zn[k] < zs[k];
            break;
        }
    }
    u32 znz = 0;
#pragma unroll
    for (int k = 0; k < S; ++k) znz |= zs[k];
    Pt<S> P;
#pragma unroll
    for (int k = 0; k < S; ++k) {
        P.X[k] = x[k];
        P.Z[k] = E.one[k];
    }
    u32 ysel[S];
#pragma unroll
    for (int k = 0; k < S; ++k) ysel[k] = (neg_smaller && znz) ? zn[k] : zs[k];
    f_mul<S>(P.Y, ysel, rr, E);                        // back to Montgomery form
    f_canon<S>(P.Y, P.Y, E);
    P.inf = 0;
    pt_store<S>(rows + j * ROW, P);
    keep[j] = ok ? 1u : 0u;
}
// idx[pos[j]] = j for the kept candidates (pos = exclusive prefix sums of keep): the gather table of the compaction
__global__ void __launch_bounds__(BLOCK) k_compact_index(u32* __restrict__ idx, const u32* __restrict__ keep, const u32* __restrict__ pos,
                                                         size_t m, size_t limit) {
    size_t j = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (j < m && keep[j] && pos[j] < limit) idx[pos[j]] = (u32)j;
}

static int prg_seed_words(const uint8_t* seed, size_t seedlen, PrgSeed& ps) {
    if (!seed || (seedlen != 32 && seedlen != 48 && seedlen != 64)) {
        set_error("PRG seed must be 32, 48 or 64 bytes (PRGHeuristic.minNoSeedBytes of SHA-256 / SHA-384 / SHA-512)");
        return VMN_ERR_UNSUPPORTED;
    }
    ps.hash = (int)seedlen * 8;
    for (size_t i = 0; i < seedlen / 4; ++i)
        ps.w[i] = ((uint32_t)seed[4 * i] << 24) | ((uint32_t)seed[4 * i + 1] << 16) | ((uint32_t)seed[4 * i + 2] << 8) | seed[4 * i + 3];
    return VMN_OK;
}
// device rows of one part of the PRG values (see k_prg_rows)
// The values wanted of a PRG stream: n consecutive ones from `first`, or the ones a device index table names.
struct PrgSel {
    size_t first = 0;
    const uint32_t* d_idx = nullptr;
};
static int prg_rows(vmn_ctx* ctx, const PrgSeed& ps, size_t n, size_t val_bytes, int val_bits, size_t part_off, size_t part_len,
                    size_t row_bytes, DevTmp& rows, const PrgSel& sel = PrgSel()) {
    DevTmp dseed(ctx);
    VMN_TRY(dseed.alloc(64));
    VMN_TRY(h2d(ctx, dseed.p, ps.w, 64));
    VMN_TRY(rows.alloc(n * row_bytes + 8));
    uint32_t top_mask = val_bits % 8 ? (1u << (val_bits % 8)) - 1 : 0xffu;
    if (ps.hash == 256)
        return launch_light(ctx, "prg", k_prg_rows<256>, grid_for(n), rows.as<uint8_t>(), row_bytes, val_bytes, top_mask, part_off, part_len,
                            (const uint32_t*)dseed.as<uint32_t>(), n, sel.first, sel.d_idx);
    if (ps.hash == 384)
        return launch_light(ctx, "prg", k_prg_rows<384>, grid_for(n), rows.as<uint8_t>(), row_bytes, val_bytes, top_mask, part_off, part_len,
                            (const uint32_t*)dseed.as<uint32_t>(), n, sel.first, sel.d_idx);
    return launch_light(ctx, "prg", k_prg_rows<512>, grid_for(n), rows.as<uint8_t>(), row_bytes, val_bytes, top_mask, part_off, part_len,
                        (const uint32_t*)dseed.as<uint32_t>(), n, sel.first, sel.d_idx);
}
// rows already on the device -> residues (mode: see k_import_be)
static int import_dev(vmn_ctx* ctx, const vmn_modulus& m, size_t nbytes, const uint8_t* d_rows, int mode, size_t n, uint32_t* d_out,
                      int* all_in_range) {
    if (all_in_range) *all_in_range = 1;
    if (n == 0) return VMN_OK;
    VMN_TRY(dev_zero(ctx, ctx->flags, sizeof(uint32_t)));
    int rc = VMN_ERR_ARG;
#define X(S_, NW_, LPE_)                                                                                               \
    if (m.S == S_)                                                                                                     \
        rc = launch(ctx, "import", k_import_be<Cfg<S_, LPE_>, NW_>, egrid(m, n), lds_bytes(m), d_out, d_rows, nbytes, nbytes, mode, n, \
                    m.d_n, m.n0inv, m.d_rr, ctx->flags);
    VMN_DISPATCH(n, X)
#undef X
    VMN_TRY(rc);
    if (all_in_range) {
        uint32_t fl = 0;
        VMN_TRY(d2h(ctx, &fl, ctx->flags, sizeof(fl)));
        *all_in_range = (fl & 1u) ? 0 : 1;
    }
    return VMN_OK;
}

// d_out[i] = (the i-th ceil(vbits / 8) bytes of PRG(seed), leading bits cleared) mod m, as a residue row.  Values that
// fit the modulus' bytes are imported directly (reduced by the import when they can reach m); wider ones -- bits(m) +
// rbitlen bits: the statistically-close-to-uniform sampling of randomElementArray -- are split as hi * 2^(8 pb) + lo
// with pb = the modulus' byte length, and recombined with one multiply-add mod m.
static int prg_residues(vmn_ctx* ctx, const vmn_modulus& m, const PrgSeed& w, size_t n, int vbits, uint32_t* d_out,
                        const PrgSel& sel = PrgSel()) {
    if (n == 0) return VMN_OK;
    const size_t Wd = elem_words(m);
    const size_t vb = ((size_t)vbits + 7) / 8, mb = ((size_t)m.nbits + 7) / 8;
    DevTmp rows(ctx);
    if (vb <= mb) {
        // one part: < 2^(8 vb) <= 2^(8 mb) <= R, reduced by the import when it can reach the modulus
        VMN_TRY(prg_rows(ctx, w, n, vb, vbits, 0, vb, vb, rows, sel));
        return import_dev(ctx, m, vb, rows.as<uint8_t>(), vbits >= m.nbits ? 2 : 0, n, d_out, nullptr);
    }
    // wider than the modulus (bits(q) + rbitlen random bits for a ring element; the 612-bit epsilon of a proof over a
    // 256-bit curve order): parts of mb bytes from the most significant end, each reduced by its import (< 2^(8 mb) <= R),
    // combined by Horner with c = 2^(8 mb) mod m:  value = (..(top c + part_1) c + ..) c + part_last  mod m
    const size_t nparts = (vb + mb - 1) / mb, top_len = vb - (nparts - 1) * mb;
    DevTmp acc(ctx), part(ctx), cdev(ctx);
    VMN_TRY(acc.alloc(n * Wd * sizeof(uint32_t)));
    VMN_TRY(part.alloc(n * Wd * sizeof(uint32_t)));
    Big c(m.NW, 0);
    c[0] = 1;
    for (size_t k = 0; k < 8 * mb; ++k) hostbig::dbl_mod(c, m.n_words);
    std::vector<uint8_t> cbe(mb);
    hostbig::to_be(c, cbe.data(), mb);
    VMN_TRY(cdev.alloc(Wd * sizeof(uint32_t)));
    int ok = 1;
    VMN_TRY(import_be(ctx, m, mb, cbe.data(), 1, cdev.as<uint32_t>(), &ok, 0, nullptr, true));      // (< m by construction: only queued)
    VMN_TRY(prg_rows(ctx, w, n, vb, vbits, 0, top_len, top_len, rows, sel));
    VMN_TRY(import_dev(ctx, m, top_len, rows.as<uint8_t>(), 2, n, acc.as<uint32_t>(), nullptr));
    uint32_t* cur = acc.as<uint32_t>();
    for (size_t pidx = 1; pidx < nparts; ++pidx) {
        VMN_TRY(prg_rows(ctx, w, n, vb, vbits, top_len + (pidx - 1) * mb, mb, mb, rows, sel));
        VMN_TRY(import_dev(ctx, m, mb, rows.as<uint8_t>(), 2, n, part.as<uint32_t>(), nullptr));
        uint32_t* dst = pidx + 1 == nparts ? d_out : cur;          // element-wise: a lane reads its operands before it writes
        int rc = VMN_ERR_ARG;                                       // dst = cur * c + part   (ring op 2 with this modulus)
#define X(S_, NW_, LPE_)                                                                                                      \
    if (m.S == S_)                                                                                                            \
        rc = launch(ctx, "ring", k_ring_elementwise<Cfg<S_, LPE_>>, egrid(m, n), lds_bytes(m), dst, (const uint32_t*)cur,          \
                    (const uint32_t*)part.as<uint32_t>(), (const uint32_t*)cdev.as<uint32_t>(), 2, n, m.d_n, m.n0inv);
        VMN_DISPATCH(n, X)
#undef X
        VMN_TRY(rc);
        cur = dst;
    }
    return VMN_OK;
}

extern "C" int vmn_prg_bytes(const uint8_t* seed, size_t seedlen, uint8_t* out, size_t nbytes) {
    ARG_CHECK(out || nbytes == 0, "null argument");
    PrgSeed ps;
    VMN_TRY(prg_seed_words(seed, seedlen, ps));
    const size_t db = (size_t)ps.digest_bytes();
    for (size_t blk = 0; blk * db < nbytes; ++blk) {
        uint8_t d256[32], d384[48], d512[64];
        const uint8_t* dg = d256;
        if (ps.hash == 256) {
            prg_block_bytes<256>(d256, ps.w, (uint32_t)blk);
        } else if (ps.hash == 384) {
            prg_block_bytes<384>(d384, ps.w, (uint32_t)blk);
            dg = d384;
        } else {
            prg_block_bytes<512>(d512, ps.w, (uint32_t)blk);
            dg = d512;
        }
        for (size_t b = 0; b < db && blk * db + b < nbytes; ++b) out[blk * db + b] = dg[b];
    }
    return VMN_OK;
}

extern "C" int vmn_random_oracle_hash(int hash_bits, const uint8_t* data, size_t len, int nout_bits, uint8_t* out) {
    ARG_CHECK((data || len == 0) && out && nout_bits > 0, "bad argument");
    ARG_CHECK(hash_bits == 256 || hash_bits == 384 || hash_bits == 512, "hash must be 256, 384 or 512 (SHA-2)");
    std::vector<uint8_t> msg(4 + len);
    msg[0] = (uint8_t)((uint32_t)nout_bits >> 24);
    msg[1] = (uint8_t)((uint32_t)nout_bits >> 16);
    msg[2] = (uint8_t)((uint32_t)nout_bits >> 8);
    msg[3] = (uint8_t)nout_bits;
    if (len) memcpy(msg.data() + 4, data, len);
    uint8_t seed[64];
    if (hash_bits == 256) {
        uint8_t d[32];
        sha256::hash(msg.data(), msg.size(), d);
        memcpy(seed, d, 32);
    } else if (hash_bits == 384) {
        sha512::hash<384>(msg.data(), msg.size(), seed);
    } else {
        sha512::hash<512>(msg.data(), msg.size(), seed);
    }
    size_t nb = ((size_t)nout_bits + 7) / 8;
    VMN_TRY(vmn_prg_bytes(seed, (size_t)hash_bits / 8, out, nb));
    if (nout_bits % 8) out[0] &= (uint8_t)((1u << (nout_bits % 8)) - 1);
    return VMN_OK;
}
extern "C" int vmn_random_oracle(const uint8_t* data, size_t len, int nout_bits, uint8_t* out) {
    return vmn_random_oracle_hash(256, data, len, nout_bits, out);
}

// values [first, first + n) of the stream, or (idx != null) the values idx[0 .. n-1]
static int rarray_from_prg_sel(vmn_group* grp, const uint8_t* seed, size_t seedlen, size_t first, const uint32_t* idx, size_t n, int bits,
                               vmn_rarray** out) {
    vmn_ctx* ctx = LANE(grp->ctx);
    PrgSeed w;
    VMN_TRY(prg_seed_words(seed, seedlen, w));
    const size_t vb = ((size_t)bits + 7) / 8;
    if (n && !idx && (first + n) > (((size_t)1 << 32) * (size_t)w.digest_bytes()) / vb) {
        set_error("PRG stream position beyond the 32-bit block counter of PRGHeuristic");
        return VMN_ERR_ARG;
    }
    vmn_rarray* a = nullptr;
    VMN_TRY(new_rarray(grp, n, &a));
    PrgSel sel;
    sel.first = first;
    DevTmp didx(ctx);
    int rc = VMN_OK;
    if (idx && n) {
        rc = didx.alloc(n * sizeof(uint32_t));
        if (rc == VMN_OK) rc = h2d(ctx, didx.p, idx, n * sizeof(uint32_t));
        sel.d_idx = didx.as<uint32_t>();
    }
    // integers that may reach the order act as field elements: reduced mod q (also when wider than q)
    if (rc == VMN_OK) rc = prg_residues(ctx, grp->Q, w, n, bits, a->d, sel);
    if (rc != VMN_OK) {
        vmn_rarray_free(a);
        return rc;
    }
    *out = a;
    return VMN_OK;
}
extern "C" int vmn_rarray_from_prg(vmn_group* grp, const uint8_t* seed, size_t seedlen, size_t n, int bits, vmn_rarray** out) {
    ARG_CHECK(grp && out && bits > 0, "bad argument");
    VMN_ENTER(LANE(grp->ctx));
    return rarray_from_prg_sel(grp, seed, seedlen, 0, nullptr, n, bits, out);
}
extern "C" int vmn_rarray_from_prg_range(vmn_group* grp, const uint8_t* seed, size_t seedlen, size_t first, size_t n, int bits,
                                         vmn_rarray** out) {
    ARG_CHECK(grp && out && bits > 0, "bad argument");
    VMN_ENTER(LANE(grp->ctx));
    return rarray_from_prg_sel(grp, seed, seedlen, first, nullptr, n, bits, out);
}
extern "C" int vmn_rarray_from_prg_gather(vmn_group* grp, const uint8_t* seed, size_t seedlen, const uint32_t* idx, size_t n, int bits,
                                          vmn_rarray** out) {
    ARG_CHECK(grp && out && bits > 0 && (idx || n == 0), "bad argument");
    VMN_ENTER(LANE(grp->ctx));
    return rarray_from_prg_sel(grp, seed, seedlen, 0, idx, n, bits, out);
}

// Random curve points from a PRG seed (see k_ec_random_points): candidates in batches, kept rows compacted in order.
static int ec_random_points(vmn_group* grp, const uint8_t* seed, size_t seedlen, size_t n, int rbitlen, vmn_garray** out) {
    vmn_ctx* ctx = LANE(grp->ctx);
    const vmn_modulus& m = grp->P;
    const vmn_curve* cv = grp->curve;
    PrgSeed ps;
    VMN_TRY(prg_seed_words(seed, seedlen, ps));
    const int pbits = hostbig::bit_length(cv->p_words);
    const int vbits = pbits + rbitlen;
    const size_t vb = ((size_t)vbits + 7) / 8, cb = ((size_t)pbits + 7) / 8;
    if (vb - cb + 1 > cb) {
        set_error("vmn_garray_from_prg: rbitlen too large for this curve");
        return VMN_ERR_UNSUPPORTED;
    }
    vmn_garray* r = nullptr;
    VMN_TRY(new_garray(grp, n, &r));
    if (n == 0) {
        *out = r;
        return VMN_OK;
    }
    const size_t Wd = elem_words(m);
    // c_m = 2^(8 cb) * R mod p as field limbs
    Big c(cv->NW, 0);
    c[0] = 1;
    for (size_t k = 0; k < 8 * cb + 28 * (size_t)cv->S; ++k) hostbig::dbl_mod(c, cv->p_words);
    std::vector<uint32_t> cl = limbs_of(c, cv->S);
    cl.resize(stride_for_limbs(cv->S), 0);
    DevTmp dc(ctx), dseed(ctx);
    int rc = dc.alloc(cl.size() * sizeof(uint32_t));
    if (rc == VMN_OK) rc = h2d(ctx, dc.p, cl.data(), cl.size() * sizeof(uint32_t));
    if (rc == VMN_OK) rc = dseed.alloc(64);
    if (rc == VMN_OK) rc = h2d(ctx, dseed.p, ps.w, 64);
    const uint32_t top_mask = vbits % 8 ? (1u << (vbits % 8)) - 1 : 0xffu;
    size_t have = 0, first = 0;                       // points kept so far, candidates consumed so far
    while (rc == VMN_OK && have < n) {
        const size_t need = n - have;
        const size_t mcand = 2 * need + need / 8 + 64;        // about half of the candidates are kept
        const size_t scan_blocks = (mcand + (size_t)BLOCK * SCAN_ITEMS - 1) / ((size_t)BLOCK * SCAN_ITEMS);
        DevTmp cand(ctx), meta(ctx);
        rc = cand.alloc(mcand * Wd * sizeof(uint32_t));
        if (rc == VMN_OK) rc = meta.alloc((3 * (mcand + 1) + scan_blocks + 8) * sizeof(uint32_t));
        if (rc != VMN_OK) break;
        uint32_t* keep = meta.as<uint32_t>();
        uint32_t* pos = keep + (mcand + 1);
        uint32_t* idx = pos + (mcand + 1);
        uint32_t* bsum = idx + (mcand + 1);
        uint32_t* total = bsum + scan_blocks;
        rc = VMN_ERR_ARG;
#define X(S_, NW_)                                                                                                            \
    if (cv->S == S_) {                                                                                                        \
        if (ps.hash == 256)                                                                                                   \
            rc = launch_light(ctx, "prg", k_ec_random_points<S_, NW_, 256>, grid_for(mcand), cand.as<uint32_t>(), keep,       \
                              (const uint32_t*)dseed.as<uint32_t>(), first, mcand, vb, top_mask, cb, (const uint32_t*)dc.as<uint32_t>(), ecdev(cv)); \
        else if (ps.hash == 384)                                                                                              \
            rc = launch_light(ctx, "prg", k_ec_random_points<S_, NW_, 384>, grid_for(mcand), cand.as<uint32_t>(), keep,       \
                              (const uint32_t*)dseed.as<uint32_t>(), first, mcand, vb, top_mask, cb, (const uint32_t*)dc.as<uint32_t>(), ecdev(cv)); \
        else                                                                                                                  \
            rc = launch_light(ctx, "prg", k_ec_random_points<S_, NW_, 512>, grid_for(mcand), cand.as<uint32_t>(), keep,       \
                              (const uint32_t*)dseed.as<uint32_t>(), first, mcand, vb, top_mask, cb, (const uint32_t*)dc.as<uint32_t>(), ecdev(cv)); \
    }
        VMN_FOR_CURVES(X)
#undef X
        if (rc != VMN_OK) break;
        // exclusive prefix sums of the keep flags, then the kept rows in candidate order
        rc = launch_light(ctx, "prg", k_u32_blocksum, (unsigned)scan_blocks, bsum, (const uint32_t*)keep, mcand);
        if (rc == VMN_OK) {
            hipLaunchKernelGGL(k_u32_scan_top, dim3(1), dim3(64), 0, ctx->stream, bsum, scan_blocks, total);
            rc = hipGetLastError() == hipSuccess ? VMN_OK : VMN_ERR_DEVICE;
        }
        if (rc == VMN_OK) rc = launch_light(ctx, "prg", k_u32_scan_apply, (unsigned)scan_blocks, pos, (uint32_t*)nullptr, (const uint32_t*)keep, (const uint32_t*)bsum, mcand);
        uint32_t kept = 0;
        if (rc == VMN_OK) rc = d2h(ctx, &kept, total, sizeof(kept));
        if (rc != VMN_OK) break;
        size_t take = std::min<size_t>(kept, need);
        // the candidates consumed: all of them when more points are needed; otherwise up to the last one taken (not needed
        // afterwards: the array is complete)
        if (take) {
            rc = launch_light(ctx, "prg", k_compact_index, grid_for(mcand), idx, (const uint32_t*)keep, (const uint32_t*)pos, mcand, take);
            int cpr = (int)(Wd / 4);
            if (rc == VMN_OK)
                rc = launch_light(ctx, "gather", k_gather, light_grid(ctx, take * cpr), reinterpret_cast<uint4*>(r->d + have * Wd),
                                  reinterpret_cast<const uint4*>(cand.as<uint32_t>()), (const uint32_t*)idx,
                                  reinterpret_cast<const uint4*>(m.d_one), take, cpr);
        }
        have += take;
        first += mcand;
    }
    if (rc != VMN_OK) {
        vmn_garray_free(r);
        return rc;
    }
    *out = r;
    return VMN_OK;
}

extern "C" int vmn_garray_from_prg(vmn_group* grp, const uint8_t* seed, size_t seedlen, size_t n, int rbitlen, vmn_garray** out) {
    ARG_CHECK(grp && out && rbitlen >= 0, "bad argument");
    vmn_ctx* ctx = LANE(grp->ctx);
    VMN_ENTER(ctx);
    if (grp->curve) return ec_random_points(grp, seed, seedlen, n, rbitlen, out);
    const vmn_modulus& m = grp->P;
    // h_i = t_i^((p - 1) / q): the cofactor is 2 for a safe-prime group (the elements are squared); any other ModPGroup
    // (demo/mixnet/group_descriptions:29, a 1024-bit p with a 256-bit q) raises to its own cofactor
    Big pm1 = m.n_words, cof, rem;
    pm1[0] -= 1;                                    // p is odd
    {
        Big qq = grp->Q.n_words;
        while (qq.size() > 1 && qq.back() == 0) qq.pop_back();
        Big qpad = qq;
        qpad.resize(std::max(qq.size(), pm1.size()), 0);
        cof = hostbig::div(pm1, qpad, &rem);
        if (!hostbig::is_zero(rem)) {
            set_error("vmn_garray_from_prg: q does not divide p - 1");
            return VMN_ERR_ARG;
        }
    }
    const bool squares = hostbig::bit_length(cof) == 2 && cof[0] == 2;
    PrgSeed w;
    VMN_TRY(prg_seed_words(seed, seedlen, w));
    const int vbits = m.nbits + rbitlen;
    vmn_garray* r = nullptr;
    VMN_TRY(new_garray(grp, n, &r));
    if (n == 0) {
        *out = r;
        return VMN_OK;
    }
    const size_t Wd = elem_words(m);
    vmn_garray* t = nullptr;
    int rc = new_garray(grp, n, &t);
    if (rc == VMN_OK) rc = prg_residues(ctx, m, w, n, vbits, t->d);      // t_i mod p
    if (rc == VMN_OK && squares) {
        rc = mul_arrays(ctx, m, t->d, t->d, Wd, n, r->d);                 // cofactor 2
    } else if (rc == VMN_OK) {
        std::vector<uint8_t> cbe(4 * cof.size());
        hostbig::to_be(cof, cbe.data(), cbe.size());
        vmn_garray_free(r);
        r = nullptr;
        rc = vmn_garray_exp_scalar(t, cbe.data(), cbe.size(), &r);
    }
    vmn_garray_free(t);
    if (rc != VMN_OK) {
        vmn_garray_free(r);
        return rc;
    }
    *out = r;
    return VMN_OK;
}

// ---- K2 fixed base ---------------------------------------------------------------------------------
// Window of a fixed-base table: ceil(ebits / w) products per exponentiation against 2^w * ceil(ebits / w) products
// to build the table, the build shared by `reuse` calls.  A base seen for the first time (the per-proof h_0) gets
// the window that is best for one call (w = 16 at N = 10^6: 2.5 GB); a base that keeps coming back (g, the public
// key) is rebuilt with the window that is best over its uses (w = 19: 17 GB at 2048 bits -- 288 GB of HBM are
// there to be used), capped per table.
static double fixed_table_cap_reuse() {
    const char* env = getenv("VMN_FIXED_TABLE_CAP");            // bytes one table of a long-lived base may take
    return env && *env ? atof(env) : 20e9;
}
static int pick_fixed_window(size_t n, int ebits, size_t row_bytes, int reuse = 1) {
    if (const char* env = getenv("VMN_FIXED_WINDOW")) {          // measurement knob: force the window
        int w = atoi(env);
        if (w >= 2 && w <= 22) return w;
    }
    if (const char* env = reuse > 1 ? getenv("VMN_FIXED_WINDOW_REUSE") : nullptr) {   // the same for long-lived bases only
        int w = atoi(env);
        if (w >= 2 && w <= 22) return w;
    }
    int best = 4;
    double best_cost = 1e300;
    const double cap = reuse > 1 ? fixed_table_cap_reuse() : 6e9;
    for (int w = 2; w <= 22; ++w) {
        int nwin = (ebits + w - 1) / w;
        double table_bytes = (double)nwin * (double)((size_t)1 << w) * (double)row_bytes;
        if (table_bytes > cap) break;
        double cost = (double)nwin * ((double)((size_t)1 << w) / (double)reuse + (double)n);
        if (cost < best_cost) {
            best_cost = cost;
            best = w;
        }
    }
    return best;
}

// Bound on the cached fixed-base tables of a group.  g and the public key stay hot; the per-proof base h_0 brings
// a new 2.5 GB table (2048 bits, w = 16) with every proof, so the least recently used tables are dropped.
static size_t fixed_cache_limit() {
    const char* env = getenv("VMN_FIXED_CACHE_BYTES");       // operational override (and the eviction test)
    if (env && *env) return (size_t)strtoull(env, nullptr, 10);
    return (size_t)64 << 30;
}

static void fixed_drop(vmn_group* g, std::map<std::string, vmn_group::FixedTable>::iterator it) {
    (void)hipStreamSynchronize(g->ctx->stream);                    // kernels reading the table may still be queued, on either lane
    if (g->ctx->helper) (void)hipStreamSynchronize(g->ctx->helper->stream);
    if (it->second.ready) (void)hipEventDestroy(it->second.ready);
    table_arena_put(g->ctx, it->second.d_tab, it->second.bytes);
    g->fixed_bytes -= it->second.bytes;
    g->fixed.erase(it);
}
static int fixed_alloc(vmn_group* g, size_t bytes, uint32_t** out) {
    if (void* kept = table_arena_take(g->ctx, bytes)) {              // a block of a dropped table: no trip to the driver
        *out = static_cast<uint32_t*>(kept);
        return VMN_OK;
    }
    for (;;) {
        bool over = g->fixed_bytes + bytes > fixed_cache_limit();
        hipError_t e = over && !g->fixed.empty() ? hipErrorOutOfMemory : hipMalloc(reinterpret_cast<void**>(out), bytes);
        if (e == hipSuccess) return VMN_OK;
        (void)hipGetLastError();
        if (g->fixed.empty()) {
            if (over && hipMalloc(reinterpret_cast<void**>(out), bytes) == hipSuccess) return VMN_OK;   // one table larger than the bound
            (void)hipGetLastError();
            table_arena_release(g->ctx);                                      // kept table blocks and
            for (vmn_ctx* c : {g->ctx, g->ctx->helper}) {                     // the cached array blocks go before the call fails
                if (!c) continue;
                std::lock_guard<std::recursive_mutex> guard__(c->mu);
                pool_release_all(c);
            }
            if (hipMalloc(reinterpret_cast<void**>(out), bytes) == hipSuccess) return VMN_OK;
            (void)hipGetLastError();
            set_error("fixed-base table allocation of %zu bytes failed", bytes);
            return VMN_ERR_NOMEM;
        }
        auto lru = g->fixed.begin();
        for (auto it = g->fixed.begin(); it != g->fixed.end(); ++it) {
            if (it->second.last_use < lru->second.last_use) lru = it;
        }
        fixed_drop(g, lru);
    }
}

// Table for (base, window) cached in the group; built on the GPU from the host squaring chain.
static int ec_normalize(vmn_ctx* ctx, const vmn_modulus& m, const uint32_t* const* ins, size_t k, size_t n, uint32_t* out);
// The squaring chain base^(2^j), j < chain, of a new table over a modular group: sequential host work (64-bit limbs, ~2 us
// per squaring at 2048 bits: 4 ms for a full-length exponent).  vmn_group_precompute_fixed runs it BEFORE it takes the
// table lock, so that the other lane's fixed-base calls are not held up behind it.
static int fixed_chain_host(const vmn_group* g, const uint8_t* base_be, size_t chain, std::vector<uint8_t>& sq_be) {
    const vmn_modulus& m = g->P;
    Big base = hostbig::from_be(base_be, g->nbytes, m.NW);
    if (hostbig::cmp(base, m.n_words) >= 0) {
        set_error("fixed base out of range");
        return VMN_ERR_FORMAT;
    }
    VMN_TRACE("fixed_table:chain_host");
    // The rows leave in the HOST's Montgomery form (x R_h mod p, R_h = 2^(64 nl)): converting each back would be a second
    // product per step of a strictly sequential chain.  One more row carries R_h^-1 mod p, and the device multiplies the
    // rows by it once they are imported (fixed_table) -- chain products on thousands of lanes instead of on one core.
    const num64::Mod& hm = *m.hm64;
    num64::Num cur = hm.to_m(num64::from_be(base_be, g->nbytes, hm.nl));
    sq_be.resize((chain + 1) * g->nbytes);
    for (size_t j = 0; j < chain; ++j) {
        num64::to_be(cur, sq_be.data() + j * g->nbytes, g->nbytes);
        hm.msqr(cur, cur);
    }
    num64::Num one(hm.nl, 0);
    one[0] = 1;
    num64::to_be(hm.from_m(one), sq_be.data() + chain * g->nbytes, g->nbytes);           // R_h^-1 mod p
    return VMN_OK;
}
static int fixed_window_for(vmn_group* g, size_t n, int ebits, int reuse_hint) {
    vmn_ctx* ctx = LANE(g->ctx);
    const vmn_modulus& m = g->P;
    // A launch over fewer elements than the chip holds lanes costs as much as a full one (the per-lane chain of
    // products is what takes the time), so small arrays are priced at the lane capacity: the window grows and the
    // chain shortens (N = 3 x 10^4: w = 12 -> 14, 171 -> 147 sequential products per exponentiation).
    // Modular groups cut the chain of a small array into up to 16 pieces on as many times the lanes (vmn_group_exp_fixed), so
    // an exponentiation of n elements is PRICED at the lanes it really occupies: max(n, lanes / pieces).  (Round 4: without
    // this a per-proof base at N = 10^4 got a 2^15-entry window -- 4.5 M products to build, 1.8 ms of the device -- to save
    // a fifth of two exponentiations of 0.55 ms.)
    size_t lanes = m.ec ? (size_t)ctx->num_cus * 4 * 64 * 2 : (size_t)ctx->num_cus * blocks_per_cu(m) * (BLOCK / m.LPE);
    if (!m.ec) {
        size_t parts = 1;
        while (parts < 16 && 2 * parts * n <= fixed_split_fill()) parts *= 2;
        lanes /= parts;
    }
    return pick_fixed_window(std::max(n, lanes), ebits, elem_words(m) * sizeof(uint32_t), reuse_hint);
}
// probe = true: *out = the cached table when it serves, nullptr when one would have to be built (nothing is built)
static int fixed_table(vmn_group* g, const uint8_t* base_be, int ebits, size_t n, vmn_group::FixedTable** out, int reuse_hint = 1,
                       bool probe = false, const std::vector<uint8_t>* chain_be = nullptr) {
    vmn_ctx* ctx = LANE(g->ctx);
    const vmn_modulus& m = g->P;
    const size_t Wd = elem_words(m);
    std::string key(reinterpret_cast<const char*>(base_be), m.ec ? 2 * g->nbytes : g->nbytes);
    int w = fixed_window_for(g, n, ebits, reuse_hint);
    int carry_uses = 1;
    *out = nullptr;
    auto it = g->fixed.find(key);
    if (it != g->fixed.end()) {
        vmn_group::FixedTable& ft = it->second;
        ft.uses += 1;
        // a base that keeps coming back earns a larger window (amortised over the uses so far, at most 16)
        // (only after many calls: a larger table is tens of milliseconds of products and gigabytes of HBM; a session that knows
        // its long-lived bases says so at setup, vmn_group_precompute_fixed)
        const int w_many = ft.uses >= 64 ? fixed_window_for(g, n, ebits, 16) : w;
        const bool grow = w_many >= ft.wbits + 2;
        if (!grow && ft.wbits >= w && ft.nwin * ft.wbits >= ebits) {
            ft.last_use = ++g->fixed_clock;
            if (ft.built_on != ctx->stream && ft.ready) VMN_HIP(hipStreamWaitEvent(ctx->stream, ft.ready, 0));   // built by the other lane
            *out = &ft;
            return VMN_OK;
        }
        if (probe) {
            ft.uses -= 1;                                // (a probe is not a use)
            return VMN_OK;
        }
        if (grow) {
            w = w_many;
            carry_uses = ft.uses;
        }
        fixed_drop(g, it);
    }
    if (probe) return VMN_OK;
    int nwin = (ebits + w - 1) / w;
    if (m.ec) {
        // doubling chain on one lane, then the same level-by-level table build with point additions
        const size_t chain = (size_t)nwin * w;
        uint32_t* d_base = nullptr;
        VMN_TRY(import_one(ctx, m, g->nbytes, base_be, &d_base));
        DevTmp sq(ctx);
        int rc = sq.alloc(chain * Wd * sizeof(uint32_t));
        vmn_group::FixedTable ft;
        ft.wbits = w;
        ft.nwin = nwin;
        ft.bytes = (size_t)nwin * ((size_t)1 << w) * Wd * sizeof(uint32_t);
        if (rc == VMN_OK) rc = fixed_alloc(g, ft.bytes, &ft.d_tab);
        if (rc == VMN_OK) {
            rc = VMN_ERR_ARG;
#define X(S_, NW_)                                                                                                \
    if (m.ec->S == S_) {                                                                                          \
        hipLaunchKernelGGL(k_ec_chain<S_>, dim3(1), dim3(64), 0, ctx->stream, sq.as<uint32_t>(),                  \
                           (const uint32_t*)d_base, (int)chain, ecdev(m.ec));                                     \
        rc = hipGetLastError() == hipSuccess ? VMN_OK : VMN_ERR_DEVICE;                                           \
    }
            VMN_FOR_CURVES(X)
#undef X
        }
        if (rc == VMN_OK)
            rc = launch_light(ctx, "fixed_table", k_fixed_seed, grid_for((size_t)nwin * (w + 1)), ft.d_tab,
                              (const uint32_t*)sq.as<uint32_t>(), w, nwin, (const uint32_t*)m.d_one, (int)Wd);
        for (int l = 1; l < w && rc == VMN_OK; ++l) {
            size_t lanes = (((size_t)1 << l) - 1) * nwin;
#define X(S_, NW_) \
    if (m.ec->S == S_) rc = note_work(ctx, m, EC_ADD * (double)lanes) ? 0 : launch_light(ctx, "fixed_table", k_ec_fixed_level<S_>, grid_for(lanes), ft.d_tab, w, nwin, l, ecdev(m.ec));
            VMN_FOR_CURVES(X)
#undef X
        }
        free_one(ctx, m, d_base);
        if (rc == VMN_OK) {                              // Z := 1 in every entry: k_ec_fixed_exp adds them with the mixed addition
            DevTmp flat(ctx);
            const size_t entries = (size_t)nwin << w;
            rc = flat.alloc(entries * Wd * sizeof(uint32_t));
            const uint32_t* whole[1] = {ft.d_tab};
            if (rc == VMN_OK) rc = ec_normalize(ctx, m, whole, 1, entries, flat.as<uint32_t>());
            if (rc == VMN_OK && hipMemcpyAsync(ft.d_tab, flat.p, entries * Wd * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
                rc = VMN_ERR_DEVICE;
        }
        if (rc != VMN_OK) {
            if (ft.d_tab) (void)hipFree(ft.d_tab);
            return rc;
        }
        ft.last_use = ++g->fixed_clock;
        ft.uses = carry_uses;
        ft.built_on = ctx->stream;
        if (hipEventCreateWithFlags(&ft.ready, hipEventDisableTiming) == hipSuccess) (void)hipEventRecord(ft.ready, ctx->stream);
        g->fixed_bytes += ft.bytes;
        auto ins = g->fixed.emplace(key, ft);
        *out = &ins.first->second;
        return VMN_OK;
    }
    // host: sq[j] = base^(2^j) mod p, j < nwin*w   (sequential chain; handed in when the caller ran it ahead of the lock)
    VMN_TRACE("fixed_table:build");
    const size_t chain = (size_t)nwin * w;
    std::vector<uint8_t> own_chain;
    if (!chain_be || chain_be->size() != (chain + 1) * g->nbytes) {
        VMN_TRY(fixed_chain_host(g, base_be, chain, own_chain));
        chain_be = &own_chain;
    }
    const std::vector<uint8_t>& sq_be = *chain_be;
    DevTmp sq(ctx);
    VMN_TRY(sq.alloc((chain + 1) * Wd * sizeof(uint32_t)));
    int ok = 1;
    VMN_TRY(import_be(ctx, m, g->nbytes, sq_be.data(), chain + 1, sq.as<uint32_t>(), &ok));
    // out of the host's Montgomery form: every row times R_h^-1 (the last row, broadcast)
    VMN_TRY(mul_arrays(ctx, m, sq.as<uint32_t>(), sq.as<uint32_t>() + chain * Wd, 0, chain, sq.as<uint32_t>()));
    vmn_group::FixedTable ft;
    ft.wbits = w;
    ft.nwin = nwin;
    ft.bytes = (size_t)nwin * ((size_t)1 << w) * Wd * sizeof(uint32_t);
    VMN_TRY(fixed_alloc(g, ft.bytes, &ft.d_tab));
    int rc = VMN_ERR_ARG;
    rc = launch_light(ctx, "fixed_table", k_fixed_seed, grid_for((size_t)nwin * (w + 1)), ft.d_tab,
                      (const uint32_t*)sq.as<uint32_t>(), w, nwin, (const uint32_t*)m.d_one, (int)Wd);
    for (int l = 1; l < w && rc == VMN_OK; ++l) {
        size_t lanes = (((size_t)1 << l) - 1) * nwin;
#define X(S_, NW_, LPE_) \
    if (m.S == S_) rc = note_work(ctx, m, (double)lanes) ? 0 : launch(ctx, "fixed_table", k_fixed_level<Cfg<S_, LPE_>>, egrid(m, lanes), lds_bytes(m), ft.d_tab, w, nwin, l, m.d_n, m.n0inv);
        VMN_DISPATCH(lanes, X)
#undef X
    }
    if (rc != VMN_OK) {
        (void)hipFree(ft.d_tab);
        return rc;
    }
    ft.last_use = ++g->fixed_clock;
    ft.uses = carry_uses;
    ft.built_on = ctx->stream;
    if (hipEventCreateWithFlags(&ft.ready, hipEventDisableTiming) == hipSuccess) (void)hipEventRecord(ft.ready, ctx->stream);
    g->fixed_bytes += ft.bytes;
    auto ins = g->fixed.emplace(key, ft);
    *out = &ins.first->second;
    return VMN_OK;
}

extern "C" int vmn_group_precompute_fixed(vmn_group* grp, const uint8_t* base_be, size_t n_hint, int uses_hint) {
    ARG_CHECK(grp && base_be && n_hint > 0, "bad argument");
    const int uses = uses_hint < 1 ? 1 : (uses_hint > 16 ? 16 : uses_hint);
    const int ebits = grp->Q.nbits;
    vmn_group::FixedTable* ft = nullptr;
    std::vector<uint8_t> chain;
    if (!grp->P.ec) {
        // Is there anything to build?  If so the squaring chain -- milliseconds of host work -- runs here, holding neither the
        // lane's lock nor the table lock: a caller that prepares the table of a per-proof base on the helper lane (the proof
        // drivers do, for h_0) must not stall other calls of that lane, nor the protocol thread's fixed-base calls, behind it.
        {
            VMN_ENTER(LANE(grp->ctx));
            std::lock_guard<std::recursive_mutex> tab_guard(grp->tab_mu);
            VMN_TRY(fixed_table(grp, base_be, ebits, n_hint, &ft, uses, true));
            if (ft) return VMN_OK;
        }
        const int w = fixed_window_for(grp, n_hint, ebits, uses);
        VMN_TRY(fixed_chain_host(grp, base_be, (size_t)((ebits + w - 1) / w) * w, chain));
    }
    VMN_ENTER(LANE(grp->ctx));
    std::lock_guard<std::recursive_mutex> tab_guard(grp->tab_mu);
    return fixed_table(grp, base_be, ebits, n_hint, &ft, uses, false, chain.empty() ? nullptr : &chain);
}

// The table of a base that will not come back (the per-proof base h_0 of a prover that is being freed): out of the group's cache,
// into the context's table arena -- where the next proof's table of the same size takes it from, so that a session of proofs
// allocates its per-proof table ONCE.  (A table that merely ages out of the cache goes the same way, fixed_drop.)  Unknown
// base: nothing happens.
extern "C" int vmn_group_release_fixed(vmn_group* grp, const uint8_t* base_be) {
    ARG_CHECK(grp && base_be, "null argument");
    VMN_ENTER(LANE(grp->ctx));
    std::lock_guard<std::recursive_mutex> tab_guard(grp->tab_mu);
    std::string key(reinterpret_cast<const char*>(base_be), grp->P.ec ? 2 * grp->nbytes : grp->nbytes);
    auto it = grp->fixed.find(key);
    if (it != grp->fixed.end()) fixed_drop(grp, it);
    return VMN_OK;
}

extern "C" int vmn_group_exp_fixed(vmn_group* grp, const uint8_t* base_be, const vmn_rarray* e, vmn_garray** out) {
    ARG_CHECK(grp && base_be && e && out, "null argument");
    ARG_CHECK(e->grp == grp, "exponent array belongs to another group");
    vmn_ctx* ctx = LANE(grp->ctx);
    VMN_ENTER(ctx);
    const size_t n = e->n;
    vmn_garray* r = nullptr;
    VMN_TRY(new_garray(grp, n, &r));
    if (n == 0) {
        *out = r;
        return VMN_OK;
    }
    int ebits = grp->Q.nbits;
    vmn_group::FixedTable* ft = nullptr;
    DevTmp ew(ctx);
    std::lock_guard<std::recursive_mutex> tab_guard(grp->tab_mu);      // until the launch is queued: the other lane may evict tables
    int rc = fixed_table(grp, base_be, ebits, n, &ft);
    if (rc == VMN_OK) rc = ew.alloc(n * (size_t)grp->Q.NW * sizeof(uint32_t));
    if (rc == VMN_OK) rc = to_words(ctx, grp->Q, e->d, n, ew.as<uint32_t>());
    if (rc == VMN_OK && grp->P.ec) {
        const vmn_modulus& m = grp->P;
        rc = VMN_ERR_ARG;
#define X(S_, NW_)                                                                                                   \
    if (m.ec->S == S_)                                                                                               \
        rc = note_work(ctx, m, EC_MADD_RUN * (double)n * (ft->nwin - 1), 0, EC_MADD * (double)n * (ft->nwin - 1)) ? 0 : launch_light(ctx, "fixed", k_ec_fixed_exp<S_>, grid_for(n), r->d, (const uint32_t*)ft->d_tab, ft->wbits, \
                          ft->nwin, (const uint32_t*)ew.as<uint32_t>(), grp->Q.NW, n, ecdev(m.ec));
        VMN_FOR_CURVES(X)
#undef X
    } else if (rc == VMN_OK) {
        // Small arrays: an element's chain of nwin - 1 dependent products is cut into `parts` pieces on `parts` times as many
        // lanes, multiplied together by a tree of element-wise products (k_fixed_exp) -- as many pieces as it takes to give
        // every SIMD about one wave of the one-lane-per-element geometry (the geometry is then picked for n x parts items).
        int parts = 1;
        const size_t fill = fixed_split_fill();
        while (parts < 16 && (size_t)(2 * parts) * n <= fill && ft->nwin / (2 * parts) >= 4) parts *= 2;
        const size_t items = n * (size_t)parts, Wd = elem_words(grp->P);
        const vmn_modulus& m = geom(ctx, grp->P, items);
        DevTmp pieces(ctx);
        if (parts > 1) rc = pieces.alloc(items * Wd * sizeof(uint32_t));
        uint32_t* dst = parts > 1 ? pieces.as<uint32_t>() : r->d;
        // one workgroup per tile rather than a persistent grid: a workgroup slot frees up every ~2 ms, so kernels of
        // the other lane (a helper's exports) are scheduled between the tiles instead of behind the whole launch
        unsigned grid = egrid(m, items);
#define X(S_, NW_, LPE_)                                                                                                 \
    if (m.S == S_)                                                                                                 \
        rc = note_work(ctx, m, (double)n * (ft->nwin - parts)) ? 0 : launch(ctx, "fixed", k_fixed_exp<Cfg<S_, LPE_>>, grid, lds_bytes(m), dst, (const uint32_t*)ft->d_tab, ft->wbits, \
                    ft->nwin, (const uint32_t*)ew.as<uint32_t>(), grp->Q.NW, n, parts, m.d_n, m.n0inv);
        if (rc == VMN_OK) {                          // (a failed allocation of the pieces must not reach the launch: dst would be null)
            rc = VMN_ERR_ARG;
            VMN_FOR_SIZES(X)
        }
#undef X
        for (int half = parts / 2; half >= 1 && rc == VMN_OK; half /= 2) {          // pieces [0, half) *= pieces [half, 2 half)
            uint32_t* lo = pieces.as<uint32_t>();
            rc = mul_arrays(ctx, grp->P, lo, lo + (size_t)half * n * Wd, Wd, (size_t)half * n, half == 1 ? r->d : lo);
        }
    }
    if (rc != VMN_OK) {
        vmn_garray_free(r);
        return rc;
    }
    *out = r;
    return VMN_OK;
}

// out (k x n rows) = the points of the k arrays with Z = 1: arrays whose rows are then ADDED INTO running sums with the mixed
// addition.  Level-by-level Montgomery trick (ec_kernels.h): chunks of K = 8 per lane; the lowest level runs per array, the
// chunk products of all k arrays are inverted together (the one Fermat chain at the top, ~0.2 ms of latency, is paid once
// per call, not once per array).  Field scratch: pref of level 0 (k n), per upper level its values, their inverses, pref.
static int ec_normalize(vmn_ctx* ctx, const vmn_modulus& m, const uint32_t* const* ins, size_t k, size_t n, uint32_t* out) {
    if (n == 0 || k == 0) return VMN_OK;
    // K values per lane and level: a launch holds n / K lanes, each a chain of K load + product steps (VMN_EC_NORMALISE_CHUNK,
    // profiles/r04_ec_normalise_chunk_sweep.txt)
    static const size_t K_env = [] {
        const char* e = getenv("VMN_EC_NORMALISE_CHUNK");
        return e && *e ? (size_t)std::max(2, atoi(e)) : (size_t)0;
    }();
    const size_t FWd = (size_t)stride_for_limbs(m.ec->S), Wd = (size_t)m.W, K = K_env ? K_env : 8, TOP = 2048;
    const size_t n1 = (n + K - 1) / K;                       // chunks (= level-1 values) per array
    std::vector<size_t> sizes{k * n, k * n1};                // values per level (level 0: the Z's of the rows, k arrays of n)
    while (sizes.back() > TOP) sizes.push_back((sizes.back() + K - 1) / K);
    const size_t L = sizes.size();
    size_t words = 0;
    std::vector<size_t> off_pref(L), off_val(L), off_inv(L);
    for (size_t l = 0; l < L; ++l) {
        off_pref[l] = words;
        words += (l + 1 < L ? sizes[l] : 0) * FWd;           // pref of level l (not for the top)
        off_val[l] = words;
        words += (l > 0 ? sizes[l] : 0) * FWd;               // values of level l
        off_inv[l] = words;
        words += (l > 0 ? sizes[l] : 0) * FWd;               // their inverses
    }
    DevTmp buf(ctx);
    VMN_TRY(buf.alloc(words * sizeof(uint32_t)));
    uint32_t* B = buf.as<uint32_t>();
    double products = 7.0 * (double)(k * n) + EC_INV * (double)sizes[L - 1];
    for (size_t l = 1; l + 1 < L; ++l) products += 3.0 * (double)sizes[l];
    int rc = VMN_ERR_ARG;
#define X(S_, NW_)                                                                                                                   \
    if (m.ec->S == S_) {                                                                                                             \
        note_work(ctx, m, products);                                                                                                 \
        rc = VMN_OK;                                                                                                                 \
        const unsigned bpa = (unsigned)((n1 + BLOCK - 1) / BLOCK);                       /* blocks per array of the two row-level launches */ \
        for (size_t a0 = 0; a0 < k && rc == VMN_OK; a0 += LEVEL_ARRAYS) {                                                            \
            const size_t ga = std::min<size_t>(LEVEL_ARRAYS, k - a0);                                                                \
            LevelInputs li{};                                                                                                        \
            for (size_t a = 0; a < ga; ++a) li.p[a] = ins[a0 + a];                                                                   \
            rc = launch_light(ctx, "normalize", k_finv_up<S_, true>, (unsigned)(bpa * ga), B + off_pref[0] + a0 * n * FWd,     \
                              B + off_val[1] + a0 * n1 * FWd, li, bpa, n, K, ecdev(m.ec));                                           \
        }                                                                                                                            \
        for (size_t l = 1; l + 1 < L && rc == VMN_OK; ++l) {                                                                         \
            LevelInputs li{};                                                                                                        \
            li.p[0] = B + off_val[l];                                                                                                \
            const unsigned bl = (unsigned)((sizes[l + 1] + BLOCK - 1) / BLOCK);                                                      \
            rc = launch_light(ctx, "normalize", k_finv_up<S_, false>, bl, B + off_pref[l], B + off_val[l + 1], li, bl,         \
                              sizes[l], K, ecdev(m.ec));                                                                             \
        }                                                                                                                            \
        if (rc == VMN_OK)                                                                                                            \
            rc = launch_light(ctx, "normalize", k_finv_top<S_>, grid_for(sizes[L - 1]), B + off_inv[L - 1],                          \
                              (const uint32_t*)(B + off_val[L - 1]), sizes[L - 1], ecdev(m.ec));                                     \
        for (size_t l = L - 2; l >= 1 && rc == VMN_OK; --l)                                                                          \
            rc = launch_light(ctx, "normalize", k_finv_down<S_>, grid_for(sizes[l + 1]), B + off_inv[l], (const uint32_t*)(B + off_inv[l + 1]), \
                              (const uint32_t*)(B + off_pref[l]), (const uint32_t*)(B + off_val[l]), sizes[l], K, ecdev(m.ec));      \
        for (size_t a0 = 0; a0 < k && rc == VMN_OK; a0 += LEVEL_ARRAYS) {                                                            \
            const size_t ga = std::min<size_t>(LEVEL_ARRAYS, k - a0);                                                                \
            LevelInputs li{};                                                                                                        \
            for (size_t a = 0; a < ga; ++a) li.p[a] = ins[a0 + a];                                                                   \
            rc = launch_light(ctx, "normalize", k_ec_normalize_down<S_>, (unsigned)(bpa * ga), out + a0 * n * Wd, li, bpa,     \
                              (const uint32_t*)(B + off_inv[1] + a0 * n1 * FWd), (const uint32_t*)(B + off_pref[0] + a0 * n * FWd),  \
                              n, K, ecdev(m.ec));                                                                                    \
        }                                                                                                                            \
    }
    VMN_FOR_CURVES(X)
#undef X
    return rc;
}

// ---- K3 multi-exponentiation -----------------------------------------------------------------------
// Window of the multi-exponentiation: per window, n bucket insertions plus the aggregation of 2^c buckets (suffix
// scan + reduction).  The weight of a bucket against an insertion was swept on the GPU (3, 6, 10, 16 at N = 10^6 /
// 4 x 10^5, 2048 / 3072 bits / P-256): flat between 3 and 10 (fewer buckets = more windows to fill), worse at 16.
static double bucket_agg_weight() {
    const char* env = getenv("VMN_BUCKET_AGG_WEIGHT");           // tuning knob
    return env && *env ? atof(env) : 3.0;
}
// curves recode the windows to signed digits (light_kernels.h: signed_digit): 2^(c-1) buckets per c-bit window, one more
// bit of the exponent to hold the last carry.  VMN_SIGNED_WINDOWS=0 turns that off (measurement).
static bool signed_windows(const vmn_modulus& m) {
    static const bool off = [] {
        const char* e = getenv("VMN_SIGNED_WINDOWS");
        return e && *e == '0';
    }();
    return m.ec != nullptr && !off;
}
static int pick_bucket_bits(size_t n, int ebits, bool ec = false, bool sgn = false) {
    if (const char* env = getenv("VMN_WINDOW_BITS")) {           // measurement knob: the window width itself
        const int c = atoi(env);
        if (c >= 2 && c <= 16) return c;
    }
    int best = 1;
    double best_cost = 1e300;
    // curves: a bucket of the aggregation (two full additions in low-occupancy scans) costs about eight insertions (mixed
    // additions at full occupancy), measured: profiles/r03_bucket_weight_sweep.txt
    // (signed windows: 24 -- the sweep of the window width itself, profiles/r03_window_bits_sweep.txt: 13 bits at 10^6 points,
    // 10 bits at 10^5; beyond 2^12 buckets per window the counting sort also leaves its LDS-privatised path)
    const double wagg = getenv("VMN_BUCKET_AGG_WEIGHT") ? bucket_agg_weight() : ec ? (sgn ? 24.0 : 8.0) : 3.0;
    for (int c = sgn ? 2 : 1; c <= (sgn ? 17 : 16); ++c) {
        int nwin = sgn ? (ebits + c) / c : (ebits + c - 1) / c;
        double cost = (double)nwin * ((double)n + wagg * (double)((size_t)1 << (sgn ? c - 1 : c)));
        if (cost < best_cost) {
            best_cost = cost;
            best = c;
        }
    }
    return best;
}

// A multi-exponentiation whose device part is queued and whose result has not been fetched yet (vmn_garray_expprod_multi_begin /
// vmn_pending_finish): what is left is a copy into pinned memory behind an event and -- for modular groups -- the Horner chain
// over the window results on the host (c * nwin squarings, sequential: ~1 ms at 2048 bits, during which a caller with other
// work for the device queues that work first).
struct vmn_pending {
    vmn_group* g = nullptr;
    vmn_ctx* lane = nullptr;
    size_t k = 0;
    int nwin = 0, c = 0;
    bool staged = false;                   // the results are on their way to lane->stage_pending[slot]; else `result` holds them
    int slot = -1;
    size_t staged_bytes = 0;
    hipEvent_t ev = nullptr;
    std::vector<uint8_t> result;
};

// Horner over the window results of k arrays (big-endian, nwin per array) -> k elements
static void horner_windows(const vmn_group* g, const uint8_t* wbe, size_t k, int nwin, int c, uint8_t* out_be) {
    VMN_TRACE("expprod:horner_host");
    const num64::Mod& hm = *g->P.hm64;
    auto horner = [&](size_t arr) {
        num64::Num acc = hm.one_m;
        for (int w = nwin - 1; w >= 0; --w) {
            for (int s2 = 0; s2 < c; ++s2) hm.mmul(acc, acc, acc);
            num64::Num ww = hm.to_m(num64::from_be(wbe + (arr * (size_t)nwin + w) * g->nbytes, g->nbytes, hm.nl));
            hm.mmul(acc, acc, ww);
        }
        num64::to_be(hm.from_m(acc), out_be + arr * g->nbytes, g->nbytes);
    };
    std::vector<std::future<void>> others;                 // one chain per array: the arrays beyond the first on threads of their own
    for (size_t arr = 1; arr < k; ++arr) {
        try {
            others.emplace_back(std::async(std::launch::async, horner, arr));
        } catch (const std::system_error&) {             // no thread to be had: this chain runs here
            horner(arr);
        }
    }
    horner(0);
    for (auto& f : others) f.get();
}

// The same over a curve: the window results arrive as affine points (x || y, the identity all 0xff), the chain of c * nwin
// doublings and nwin additions runs in Jacobian coordinates on 64-bit limbs (csrc/hostcurve.h) -- ~0.3 ms at P-256, where the
// one GPU lane per array of k_ec_horner took 1.3 ms with the device otherwise empty.
static void horner_windows_ec(const vmn_group* g, const uint8_t* wbe, size_t k, int nwin, int c, uint8_t* out_be) {
    VMN_TRACE("expprod:horner_host");
    num64::HostCurve hc = g->P.ec->host;
    hc.cb = g->nbytes;                                   // the wire width of a coordinate (may be the Java width)
    const size_t pb = 2 * g->nbytes;
    auto horner = [&](size_t arr) {
        num64::HostCurve::Jac acc;
        for (int w = nwin - 1; w >= 0; --w) {
            for (int s2 = 0; s2 < c; ++s2) acc = hc.dbl(acc);
            const uint8_t* pt = wbe + (arr * (size_t)nwin + w) * pb;
            num64::Num x, y;
            if (hc.decode(num64::Bytes(pt, pt + pb), x, y)) acc = hc.add_affine(acc, x, y);
        }
        num64::Bytes enc = hc.encode(acc);
        memcpy(out_be + arr * pb, enc.data(), pb);
    };
    std::vector<std::future<void>> others;
    for (size_t arr = 1; arr < k; ++arr) {
        try {
            others.emplace_back(std::async(std::launch::async, horner, arr));
        } catch (const std::system_error&) {
            horner(arr);
        }
    }
    horner(0);
    for (auto& f : others) f.get();
}

// a landing buffer of the lane for a pending result, if one is free (PENDING_SLOTS multi-exponentiations in flight per lane)
static uint8_t* claim_pending_stage(vmn_ctx* ctx, size_t bytes, vmn_pending* pend) {
    int k = -1;
    for (int i = 0; i < vmn_ctx::PENDING_SLOTS && k < 0; ++i)
        if (!ctx->stage_pending_busy[i]) k = i;
    if (k < 0) return nullptr;
    if (ctx->stage_pending_bytes[k] < bytes) {
        if (ctx->stage_pending[k]) (void)hipHostFree(ctx->stage_pending[k]);
        ctx->stage_pending[k] = nullptr;
        ctx->stage_pending_bytes[k] = 0;
        const size_t want = std::max<size_t>(bytes, (size_t)1 << 18);
        if (hipHostMalloc(&ctx->stage_pending[k], want, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            ctx->stage_pending[k] = nullptr;
            return nullptr;
        }
        ctx->stage_pending_bytes[k] = want;
    }
    ctx->stage_pending_busy[k] = true;
    pend->slot = k;
    return static_cast<uint8_t*>(ctx->stage_pending[k]);
}

// queue export + copy of `rows` rows into the claimed landing buffer and the event that says they have arrived
static int stage_pending(vmn_ctx* ctx, const vmn_modulus& m, size_t nbytes, const uint32_t* d_rows, size_t rows, uint8_t* land,
                         vmn_pending* pend) {
    int rc = export_be(ctx, m, nbytes, d_rows, rows, land, 0, true);
    if (rc == VMN_OK && hipEventCreateWithFlags(&pend->ev, hipEventDisableTiming) != hipSuccess) rc = VMN_ERR_DEVICE;
    if (rc == VMN_OK && hipEventRecord(pend->ev, ctx->stream) != hipSuccess) rc = VMN_ERR_DEVICE;
    if (rc != VMN_OK) {
        if (rc == VMN_ERR_DEVICE) set_error("multi-exponentiation: recording the completion event failed");
        (void)hipStreamSynchronize(ctx->stream);          // the copy may be queued: the buffer is free again once it has run
        if (pend->ev) (void)hipEventDestroy(pend->ev);
        pend->ev = nullptr;
        ctx->stage_pending_busy[pend->slot] = false;
        return rc;
    }
    pend->staged = true;
    return VMN_OK;
}

// prod_i x[i]^e[i] with packed-word exponents on the device -> big-endian element on the host
// k arrays with the SAME exponents (the 2*width components of a ciphertext array, or u / h / w' under one
// batching vector): the counting sort of the exponent digits and the shape of the product trees are computed
// once and reused for every array; out_be receives k elements.
// (pend: queue the device part only and leave the rest to vmn_pending_finish; out_be is then not written)
static int expprod_words(vmn_group* g, const uint32_t* const* xs, size_t k, const uint32_t* e_words, int ewords,
                         int ebits, size_t n, uint8_t* out_be, vmn_pending* pend = nullptr) {
    vmn_ctx* ctx = LANE(g->ctx);
    const vmn_modulus& m = g->P;
    const size_t Wd = elem_words(m);
    const size_t ebytes_out = m.ec ? 2 * g->nbytes : g->nbytes;
    std::vector<uint8_t> direct;                       // pend: results that are known at once
    if (pend) {
        pend->g = g;
        pend->lane = ctx;
        pend->k = k;
        if (n == 0) {
            direct.resize(k * ebytes_out);
            out_be = direct.data();
        }
    }
    if (n == 0) {
        for (size_t a = 0; a < k; ++a) {
            uint8_t* o = out_be + a * ebytes_out;
            if (m.ec) {
                memset(o, 0xff, ebytes_out);          // the identity: point at infinity
            } else {
                Big one(m.NW, 0);
                one[0] = 1;
                hostbig::to_be(one, o, g->nbytes);
            }
        }
        if (pend) pend->result = std::move(direct);
        return VMN_OK;
    }
    if (ebits < 1) ebits = 1;
    const bool sgn = signed_windows(m);
    if (sgn) ebits = 32 * ewords;                      // everything the words can hold: the recoding must not lose a carry
    if (m.ec && n >= ((size_t)1 << 31)) {
        set_error("multi-exponentiation over a curve: at most 2^31 - 1 points per array");
        return VMN_ERR_UNSUPPORTED;
    }
    const int c = pick_bucket_bits(n, ebits, m.ec != nullptr, sgn);
    const int nwin = sgn ? (ebits + c) / c : (ebits + c - 1) / c;     // signed: ebits + 1 bits
    const int cb = sgn ? c - 1 : c;                    // bucket bits of a window; bucket j of a signed window holds the digits +-(j + 1)
    const size_t nb = (size_t)1 << cb;
    const size_t nbuckets = (size_t)nwin * nb;
    // fan-in of the per-bucket product tree.  A lane sums one chunk of at most F items; the lanes of a wave wait for the longest
    // chunk, so F is best a little above the typical bucket size n / 2^c (most buckets are then ONE chunk, the wave's longest
    // chunk is close to its average, and hardly anything is left for the upper levels -- which run full additions on few lanes).
    static const uint32_t F_env = [] {
        const char* e = getenv("VMN_TREE_FANIN");
        return e && *e ? (uint32_t)std::max(4, atoi(e)) : 0u;
    }();
    // Curves: 16 -- the sum of a chunk runs in XYZZ registers and is turned into a Jacobian row once per chunk (2 products),
    // and the upper levels add Jacobian rows at 11M + 5S: both favour longer chunks (profiles/r04_ec_fanin_sweep_xyzz.txt).
    const uint32_t F = F_env ? F_env : (m.ec ? 16 : 8);
    DevTmp meta(ctx), sorted(ctx), buckets(ctx), wres(ctx), itemsA(ctx), itemsB(ctx);
    // u32 scratch: counts[nb] | cursor[nb] | off0[nb+1] | per tree level l < LV: cnt_l[nb], off_l[nb+1] | bsum | misc.
    // The shape of the product trees depends on the exponents only: the per-level counts and offsets are computed once,
    // before any array goes through the levels (below).
    const int LV = 12;                               // fan-in 8: 8^12 items per bucket
    const size_t scan_blocks = (nbuckets + (size_t)BLOCK * SCAN_ITEMS - 1) / ((size_t)BLOCK * SCAN_ITEMS);
    VMN_TRY(meta.alloc(((3 + 2 * LV) * (nbuckets + 1) + scan_blocks + 2 * LV + 16) * sizeof(uint32_t)));
    uint32_t* counts = meta.as<uint32_t>();
    uint32_t* cursor = counts + (nbuckets + 1);
    uint32_t* off0 = cursor + (nbuckets + 1);
    uint32_t* levels0 = off0 + (nbuckets + 1);
    auto cnt_of = [&](int level) { return levels0 + (size_t)(2 * level) * (nbuckets + 1); };
    auto off_of = [&](int level) { return levels0 + (size_t)(2 * level + 1) * (nbuckets + 1); };
    uint32_t* bsum = levels0 + (size_t)(2 * LV) * (nbuckets + 1);
    uint32_t* misc = bsum + scan_blocks;             // per tree level: [2 l] = total, [2 l + 1] = max count
    VMN_TRY(sorted.alloc((size_t)nwin * n * sizeof(uint32_t)));
    // The bucket aggregation (suffix scan + reduction over nwin x 2^c rows per array) is a handful of short launches: it
    // runs once for a GROUP of arrays (all k when their buckets fit 16 GB), not once per array -- over curves, where a
    // multi-exponentiation has 7 arrays and few buckets, that is a quarter of the kernel time of a proof.
    const size_t bucket_bytes = nbuckets * Wd * sizeof(uint32_t);
    const size_t G = std::max<size_t>(1, std::min<size_t>(k, ((size_t)16 << 30) / (2 * bucket_bytes)));
    VMN_TRY(buckets.alloc(2 * G * bucket_bytes));
    VMN_TRY(wres.alloc(k * (size_t)nwin * Wd * sizeof(uint32_t)));      // window results of all k arrays
    uint32_t* Ball = buckets.as<uint32_t>();
    uint32_t* Ssuf = Ball + G * nbuckets * Wd;
    auto scan_u32 = [&](uint32_t* out, uint32_t* out2, const uint32_t* in, uint32_t* total) -> int {
        VMN_TRY(launch_light(ctx, "expprod_sort", k_u32_blocksum, (unsigned)scan_blocks, bsum, in, nbuckets));
        hipLaunchKernelGGL(k_u32_scan_top, dim3(1), dim3(64), 0, ctx->stream, bsum, scan_blocks, total);
        VMN_HIP(hipGetLastError());
        return launch_light(ctx, "expprod_sort", k_u32_scan_apply, (unsigned)scan_blocks, out, out2, in, (const uint32_t*)bsum, nbuckets);
    };
    // counting sort of (window, digit)
    {
    VMN_TRACE("expprod:sort");
    VMN_TRY(dev_zero(ctx, counts, nbuckets * sizeof(uint32_t)));
    const unsigned gx = std::max<unsigned>(1, std::min<unsigned>((unsigned)((n + BLOCK - 1) / BLOCK), (unsigned)(ctx->num_cus * 8 / std::max(nwin, 1) + 1)));
    if (sgn) {
        VMN_TRY(launch_light(ctx, "expprod_sort", k_bucket_hist<true>, gx * (unsigned)nwin, counts, e_words, ewords, n, c, nwin, gx, ebits));
        VMN_TRY(scan_u32(off0, cursor, counts, misc + 2 * LV));
        VMN_TRY(launch_light(ctx, "expprod_sort", k_bucket_scatter<true>, gx * (unsigned)nwin, sorted.as<uint32_t>(), cursor,
                             e_words, ewords, n, c, nwin, gx, ebits));           // (zero digits were never inserted)
    } else {
        VMN_TRY(launch_light(ctx, "expprod_sort", k_bucket_hist<false>, gx * (unsigned)nwin, counts, e_words, ewords, n, c, nwin, gx, ebits));
        VMN_TRY(scan_u32(off0, cursor, counts, misc + 2 * LV));
        VMN_TRY(launch_light(ctx, "expprod_sort", k_bucket_scatter<false>, gx * (unsigned)nwin, sorted.as<uint32_t>(), cursor,
                             e_words, ewords, n, c, nwin, gx, ebits));
        VMN_TRY(launch_light(ctx, "expprod_sort", k_bucket_drop_zero, grid_for(nwin), counts, c, nwin));
    }
    }
    // per-bucket product tree with fan-in F, level by level until every bucket holds <= 1 item.  The SHAPE of the trees
    // (items per bucket and level, offsets, totals) depends on the exponents only: it is computed first, once; then the
    // arrays go through the levels in groups -- one launch per level for up to LEVEL_ARRAYS arrays (k_bucket_level).
    std::vector<std::pair<uint32_t, uint32_t>> shape;             // (total items, max per bucket) per level
    {
        VMN_TRACE("expprod:shape");
        // Two round trips to the host, not one per level: the largest bucket after level 0 says how many levels there are
        // (the largest bucket of level l + 1 is ceil(largest of level l / F)); they are then queued together and their totals
        // come back in one copy.
        const uint32_t* cnt_in = counts;
        uint32_t hm[2 * LV] = {0};
        VMN_TRY(dev_zero(ctx, misc, 2 * LV * sizeof(uint32_t)));
        auto queue_level = [&](int level) -> int {
            VMN_TRY(launch_light(ctx, "expprod_sort", k_task_counts, grid_for(nbuckets), cnt_of(level), cnt_in, nbuckets, F, misc + 2 * level + 1));
            VMN_TRY(scan_u32(off_of(level), (uint32_t*)nullptr, cnt_of(level), misc + 2 * level));
            cnt_in = cnt_of(level);
            return VMN_OK;
        };
        VMN_TRY(queue_level(0));
        VMN_TRY(d2h(ctx, hm, misc, 2 * sizeof(uint32_t)));
        int levels = 1;
        for (uint32_t mx = hm[1]; mx > 1; mx = (mx + F - 1) / F) {
            if (levels == LV) {
                set_error("multi-exponentiation: a bucket holds more than %u^%d items", F, LV);
                return VMN_ERR_UNSUPPORTED;
            }
            VMN_TRY(queue_level(levels++));
        }
        if (levels > 1) {
            VMN_TRY(d2h(ctx, hm + 2, misc + 2, 2 * (levels - 1) * sizeof(uint32_t)));
        }
        for (int level = 0; level < levels; ++level) shape.emplace_back(hm[2 * level], hm[2 * level + 1]);
        if (shape.back().second > 1) {
            set_error("multi-exponentiation: the bucket trees did not close after %d levels", levels);
            return VMN_ERR_DEVICE;
        }
    }
    DevTmp normalised(ctx);                              // curves: the k arrays with Z = 1, so that the first level's additions are mixed
    // ... unless the call is small: normalising is a fixed chain of ~10 short launches with a Fermat power at its top (~0.3 ms
    // whatever the size), the full additions it saves cost k n nwin x ~1100 instructions.  VMN_EC_NORMALISE_MIN moves the bound.
    const char* nm_env = getenv("VMN_EC_NORMALISE_MIN");        // (read per call: the tests run both first levels at their sizes)
    const size_t normalise_min = nm_env && *nm_env ? (size_t)strtoull(nm_env, nullptr, 10) : (size_t)131072;
    const bool ec_rows_normalised = m.ec && k * n >= normalise_min;
    if (ec_rows_normalised) {
        VMN_TRACE("expprod:normalise");
        VMN_TRY(normalised.alloc(k * n * Wd * sizeof(uint32_t)));
        VMN_TRY(ec_normalize(ctx, m, xs, k, n, normalised.as<uint32_t>()));
    }
    // arrays per launch: as many as the aggregation group holds, the kernel takes and the level buffers allow (24 GB)
    const size_t cap0 = (size_t)shape[0].first + 1, cap1 = shape.size() > 1 ? (size_t)shape[1].first + 1 : 1;
    const size_t row_bytes = Wd * sizeof(uint32_t);
    const size_t by_mem = std::max<size_t>(1, ((size_t)24 << 30) / ((cap0 + cap1) * row_bytes));
    const size_t GL = std::max<size_t>(1, std::min<size_t>({G, (size_t)LEVEL_ARRAYS, by_mem}));
    VMN_TRY(itemsA.alloc(GL * cap0 * row_bytes));
    VMN_TRY(itemsB.alloc(GL * cap1 * row_bytes));
    for (size_t arr0 = 0; arr0 < k; arr0 += GL) {
        VMN_TRACE("expprod:levels_and_aggregation");
        const size_t gl = std::min(GL, k - arr0);
        const uint32_t* cnt_in = counts;
        const uint32_t* off_in = off0;
        LevelInputs ins{};
        for (size_t a = 0; a < gl; ++a) ins.p[a] = ec_rows_normalised ? normalised.as<uint32_t>() + (arr0 + a) * n * Wd : xs[arr0 + a];
        uint32_t* items_out = itemsA.as<uint32_t>();
        uint32_t* items_other = itemsB.as<uint32_t>();
        size_t out_cap = cap0, other_cap = cap1;
        bool first = true;
        int rc = VMN_OK;
        for (size_t level = 0; level < shape.size(); ++level) {
            const size_t total_out = shape[level].first;
            const double level_in = level == 0 ? (double)nwin * (double)n : (double)shape[level - 1].first;
            const double level_products = std::max(0.0, level_in - (double)total_out) * (double)gl;
            const size_t out_stride = out_cap * Wd;
            if (total_out > 0 && m.ec) {
                rc = VMN_ERR_ARG;
                const unsigned bpa = grid_for(total_out);
#define X(S_, NW_)                                                                                                     \
    if (m.ec->S == S_) {                                                                                               \
        rc = note_work(ctx, m, (first && ec_rows_normalised ? EC_MADD_RUN : EC_ADD) * level_products, 0,                        \
                       (first && ec_rows_normalised ? EC_MADD : EC_ADD) * level_products) ? 0                                  \
             : first && !ec_rows_normalised ? launch_light(ctx, "expprod", k_ec_bucket_first_jacobian<S_>, bpa * (unsigned)gl, items_out, out_stride, ins, bpa, \
                                  (const uint32_t*)sorted.as<uint32_t>(), off_in, cnt_in, (const uint32_t*)off_of((int)level),    \
                                  nbuckets, total_out, F, ecdev(m.ec))                                                 \
             : first ? launch_light(ctx, "expprod", k_ec_bucket_level<S_, true>, bpa * (unsigned)gl, items_out, out_stride, ins, bpa, \
                                  (const uint32_t*)sorted.as<uint32_t>(), off_in, cnt_in, (const uint32_t*)off_of((int)level),    \
                                  nbuckets, total_out, F, ecdev(m.ec))                                                 \
                   : launch_light(ctx, "expprod", k_ec_bucket_level<S_, false>, bpa * (unsigned)gl, items_out, out_stride, ins, bpa,        \
                                  (const uint32_t*)nullptr, off_in, cnt_in, (const uint32_t*)off_of((int)level),        \
                                  nbuckets, total_out, F, ecdev(m.ec));                                                \
    }
                VMN_FOR_CURVES(X)
#undef X
                VMN_TRY(rc);
            } else if (total_out > 0) {
                rc = VMN_ERR_ARG;
#define X(S_, NW_, LPE_)                                                                                                      \
    if (m.S == S_) {                                                                                                    \
        const unsigned bpa = egrid(m, total_out);                                                                       \
        rc = note_work(ctx, m, level_products) ? 0 : first ? launch(ctx, "expprod", k_bucket_level<Cfg<S_, LPE_>, true>, bpa * (unsigned)gl, lds_bytes(m), items_out, out_stride,    \
                            ins, bpa, (const uint32_t*)sorted.as<uint32_t>(), off_in, cnt_in, (const uint32_t*)off_of((int)level), \
                            nbuckets, total_out, F, m.d_n, m.n0inv)                                                     \
                   : launch(ctx, "expprod", k_bucket_level<Cfg<S_, LPE_>, false>, bpa * (unsigned)gl, lds_bytes(m), items_out, out_stride,   \
                            ins, bpa, (const uint32_t*)nullptr, off_in, cnt_in, (const uint32_t*)off_of((int)level), nbuckets,     \
                            total_out, F, m.d_n, m.n0inv);                                                              \
    }
                VMN_DISPATCH(total_out * gl, X)
#undef X
                VMN_TRY(rc);
            }
            // the outputs become the next level's inputs
            first = false;
            for (size_t a = 0; a < gl; ++a) ins.p[a] = items_out + a * out_stride;
            cnt_in = cnt_of((int)level);
            off_in = off_of((int)level);
            std::swap(items_out, items_other);
            std::swap(out_cap, other_cap);
        }
        for (size_t a = 0; a < gl; ++a) {
            const size_t arr = arr0 + a;
            uint32_t* B = Ball + (arr % G) * nbuckets * Wd;
            VMN_TRY(launch_light(ctx, "expprod_agg", k_bucket_finalize, light_grid(ctx, nbuckets * (Wd / 4)),
                                 reinterpret_cast<uint4*>(B), reinterpret_cast<const uint4*>(ins.p[a]), off_in, cnt_in,
                                 nbuckets, reinterpret_cast<const uint4*>(m.d_one), (int)(Wd / 4)));
            if (arr % G == G - 1 || arr + 1 == k) {
                // W_win = prod_{d>=1} B[d]^d = prod_{d>=1} (suffix product S_d): suffix scan, blank d = 0, reduce -- for the group
                const size_t gsz = arr % G + 1, first_arr = arr + 1 - gsz, segs = gsz * (size_t)nwin;
                VMN_TRY(scan_affine(ctx, m, Ball, nullptr, segs * nb, nb, 1, Ssuf));
                if (!sgn)                              // (signed windows: bucket 0 holds the digits +-1 and counts)
                    VMN_TRY(launch_light(ctx, "expprod_agg", k_set_segment_heads, grid_for(segs * (Wd / 4)), reinterpret_cast<uint4*>(Ssuf),
                                         nb, segs, reinterpret_cast<const uint4*>(m.d_one), (int)(Wd / 4)));
                VMN_TRY(reduce_segments(ctx, m, Ssuf, nb, segs, true, wres.as<uint32_t>() + first_arr * (size_t)nwin * Wd));
            }
        }
    }
    // Horner over the windows, once for all k arrays: the chain of c * nwin doublings / squarings is sequential
    // (one lane per array on the GPU for curves; on the host for modular groups), so it is done for the k arrays
    // together and the k results leave in one copy.
    if (pend) {
        pend->nwin = nwin;
        pend->c = c;
    }
    static const bool horner_on_device = [] {          // measurement knob: the one-lane-per-array kernel of round 2
        const char* e = getenv("VMN_EC_HORNER_DEVICE");
        return e && *e == '1';
    }();
    if (m.ec && horner_on_device) {
        VMN_TRACE("expprod:horner_device");
        DevTmp res(ctx);
        VMN_TRY(res.alloc(k * Wd * sizeof(uint32_t)));
        int rc = VMN_ERR_ARG;
#define X(S_, NW_)                                                                                                        \
    if (m.ec->S == S_)                                                                                                    \
        rc = launch_light(ctx, "expprod_agg", k_ec_horner<S_>, grid_for(k), res.as<uint32_t>(), (const uint32_t*)wres.as<uint32_t>(), \
                          nwin, c, (int)k, ecdev(m.ec));
        VMN_FOR_CURVES(X)
#undef X
        VMN_TRY(rc);
        if (pend) {
            pend->nwin = 0;                            // (finish: the staged bytes ARE the results)
            pend->staged_bytes = k * ebytes_out;
            if (uint8_t* land = claim_pending_stage(ctx, pend->staged_bytes, pend))
                return stage_pending(ctx, m, g->nbytes, res.as<uint32_t>(), k, land, pend);
            pend->result.resize(k * ebytes_out);
            out_be = pend->result.data();
        }
        return export_be(ctx, m, g->nbytes, res.as<uint32_t>(), k, out_be);
    }
    // the window results leave the device (curves: as affine points) and the chain over them runs on the host
    const size_t wbytes = k * (size_t)nwin * ebytes_out;
    if (pend) {
        pend->staged_bytes = wbytes;
        if (uint8_t* land = claim_pending_stage(ctx, wbytes, pend))
            return stage_pending(ctx, m, g->nbytes, wres.as<uint32_t>(), k * (size_t)nwin, land, pend);
        pend->result.resize(k * ebytes_out);
        out_be = pend->result.data();
    }
    std::vector<uint8_t> wbe(wbytes);
    {
        VMN_TRACE("expprod:export_windows");
        VMN_TRY(export_be(ctx, m, g->nbytes, wres.as<uint32_t>(), k * (size_t)nwin, wbe.data()));
    }
    if (m.ec) horner_windows_ec(g, wbe.data(), k, nwin, c, out_be);
    else horner_windows(g, wbe.data(), k, nwin, c, out_be);
    return VMN_OK;
}

extern "C" int vmn_garray_expprod(const vmn_garray* x, const vmn_rarray* e, int ebits, uint8_t* out_be) {
    ARG_CHECK(x && e && out_be, "null argument");
    ARG_CHECK(x->grp == e->grp && x->n == e->n, "arrays differ in group or size");
    vmn_group* g = x->grp;
    vmn_ctx* ctx = LANE(g->ctx);
    VMN_ENTER(ctx);
    if (ebits <= 0 || ebits > g->Q.nbits) ebits = g->Q.nbits;
    DevTmp ew(ctx);
    VMN_TRY(ew.alloc(std::max<size_t>(x->n, 1) * (size_t)g->Q.NW * sizeof(uint32_t)));
    VMN_TRY(to_words(ctx, g->Q, e->d, e->n, ew.as<uint32_t>()));
    const uint32_t* xs[1] = {x->d};
    return expprod_words(g, xs, 1, ew.as<uint32_t>(), g->Q.NW, ebits, x->n, out_be);
}
extern "C" int vmn_garray_expprod_ints(const vmn_garray* x, const uint8_t* exps_be, size_t ebytes, int ebits, uint8_t* out_be) {
    ARG_CHECK(x && out_be && (exps_be || x->n == 0) && ebytes > 0, "null argument");
    vmn_group* g = x->grp;
    vmn_ctx* ctx = LANE(g->ctx);
    VMN_ENTER(ctx);
    if (ebits <= 0 || (size_t)ebits > 8 * ebytes) ebits = (int)(8 * ebytes);
    int ewords = (ebits + 31) / 32;
    std::vector<uint32_t> hw;
    be_ints_to_words(exps_be, ebytes, x->n, ewords, hw);
    DevTmp ew(ctx);
    VMN_TRY(ew.alloc(std::max<size_t>(hw.size(), 1) * sizeof(uint32_t)));
    VMN_TRY(h2d(ctx, ew.p, hw.data(), hw.size() * sizeof(uint32_t)));
    const uint32_t* xs[1] = {x->d};
    return expprod_words(g, xs, 1, ew.as<uint32_t>(), ewords, ebits, x->n, out_be);
}

extern "C" int vmn_garray_expprod_multi(const vmn_garray* const* xs, size_t k, const vmn_rarray* e, int ebits, uint8_t* out_be) {
    ARG_CHECK(xs && k > 0 && e && out_be, "null argument");
    vmn_group* g = e->grp;
    for (size_t a = 0; a < k; ++a) ARG_CHECK(xs[a] && xs[a]->grp == g && xs[a]->n == e->n, "arrays differ in group or size");
    vmn_ctx* ctx = LANE(g->ctx);
    VMN_ENTER(ctx);
    if (ebits <= 0 || ebits > g->Q.nbits) ebits = g->Q.nbits;
    DevTmp ew(ctx);
    VMN_TRY(ew.alloc(std::max<size_t>(e->n, 1) * (size_t)g->Q.NW * sizeof(uint32_t)));
    VMN_TRY(to_words(ctx, g->Q, e->d, e->n, ew.as<uint32_t>()));
    std::vector<const uint32_t*> ptrs(k);
    for (size_t a = 0; a < k; ++a) ptrs[a] = xs[a]->d;
    return expprod_words(g, ptrs.data(), k, ew.as<uint32_t>(), g->Q.NW, ebits, e->n, out_be);
}

// The same in two halves: _begin queues the device part and returns; vmn_pending_finish waits for it and completes the result
// (modular groups: the Horner chain over the windows, on the host).  Between the two the caller queues whatever else it has for
// the device -- a proof's fixed-base powers run while the host squares.  Two multi-exponentiations per lane can be in flight;
// a third _begin computes its result at once (finish then only hands it over).
extern "C" int vmn_garray_expprod_multi_begin(const vmn_garray* const* xs, size_t k, const vmn_rarray* e, int ebits, vmn_pending** out) {
    ARG_CHECK(xs && k > 0 && e && out, "null argument");
    vmn_group* g = e->grp;
    for (size_t a = 0; a < k; ++a) ARG_CHECK(xs[a] && xs[a]->grp == g && xs[a]->n == e->n, "arrays differ in group or size");
    vmn_ctx* ctx = LANE(g->ctx);
    VMN_ENTER(ctx);
    if (ebits <= 0 || ebits > g->Q.nbits) ebits = g->Q.nbits;
    DevTmp ew(ctx);
    VMN_TRY(ew.alloc(std::max<size_t>(e->n, 1) * (size_t)g->Q.NW * sizeof(uint32_t)));
    VMN_TRY(to_words(ctx, g->Q, e->d, e->n, ew.as<uint32_t>()));
    std::vector<const uint32_t*> ptrs(k);
    for (size_t a = 0; a < k; ++a) ptrs[a] = xs[a]->d;
    std::unique_ptr<vmn_pending> p(new vmn_pending());
    VMN_TRY(expprod_words(g, ptrs.data(), k, ew.as<uint32_t>(), g->Q.NW, ebits, e->n, nullptr, p.get()));
    *out = p.release();
    return VMN_OK;
}
static void pending_release(vmn_pending* p) {
    if (p->staged) {
        std::lock_guard<std::recursive_mutex> guard__(p->lane->mu);
        (void)hipSetDevice(p->lane->device);
        if (p->ev) (void)hipEventSynchronize(p->ev);      // the copy into the landing buffer must not outlive the claim
        p->lane->stage_pending_busy[p->slot] = false;
        p->staged = false;
    }
    if (p->ev) (void)hipEventDestroy(p->ev);
    p->ev = nullptr;
}
extern "C" int vmn_pending_finish(vmn_pending* p, uint8_t* out_be) {
    ARG_CHECK(p && out_be, "null argument");
    std::unique_ptr<vmn_pending> own(p);
    const size_t ebytes_out = p->g->P.ec ? 2 * p->g->nbytes : p->g->nbytes;
    if (!p->staged) {
        memcpy(out_be, p->result.data(), p->k * ebytes_out);
        return VMN_OK;
    }
    std::vector<uint8_t> landed(p->staged_bytes);
    {
        VMN_TRACE("expprod:wait_pending");
        std::lock_guard<std::recursive_mutex> guard__(p->lane->mu);
        hipError_t he = hipSetDevice(p->lane->device);
        if (he == hipSuccess) he = hipEventSynchronize(p->ev);
        if (he == hipSuccess) memcpy(landed.data(), p->lane->stage_pending[p->slot], p->staged_bytes);
        p->lane->stage_pending_busy[p->slot] = false;
        p->staged = false;
        (void)hipEventDestroy(p->ev);
        p->ev = nullptr;
        if (he != hipSuccess) {
            set_error("vmn_pending_finish: waiting for the device failed: %s", hipGetErrorString(he));
            return VMN_ERR_DEVICE;
        }
    }
    if (p->nwin == 0) memcpy(out_be, landed.data(), p->staged_bytes);            // (the device did the chain: VMN_EC_HORNER_DEVICE)
    else if (p->g->P.ec) horner_windows_ec(p->g, landed.data(), p->k, p->nwin, p->c, out_be);
    else horner_windows(p->g, landed.data(), p->k, p->nwin, p->c, out_be);
    return VMN_OK;
}
extern "C" size_t vmn_pending_bytes(const vmn_pending* p) { return p ? p->k * (p->g->P.ec ? 2 * p->g->nbytes : p->g->nbytes) : 0; }
extern "C" void vmn_pending_free(vmn_pending* p) {
    if (!p) return;
    pending_release(p);
    delete p;
}

// ---- membership, partials ----------------------------------------------------------------------------
extern "C" int vmn_garray_is_member(const vmn_garray* x, int* all_members) {
    ARG_CHECK(x && all_members, "null argument");
    vmn_group* g = x->grp;
    vmn_ctx* ctx = LANE(g->ctx);
    VMN_ENTER(ctx);
    *all_members = 1;
    if (x->n == 0) return VMN_OK;
    if (g->P.ec) return VMN_OK;      // prime-order curve: every point that passed the import's curve check is a member
    const vmn_modulus& m = g->P;
    const size_t Wd = elem_words(m);
    // safe-prime group: the Jacobi symbol decides (a tenth of the work of x^q = 1) -- one element per lane up to 2048 bits,
    // the element's own two / four lanes at 3072 / 4096 bits (one lane holding all 110 limbs of a and m runs at a third
    // of the speed: 220 live registers)
    if (!getenv("VMN_MEMBER_BY_POWER")) {
        Big twoq = g->Q.n_words;
        hostbig::dbl_mod(twoq, m.n_words);
        Big pm1 = m.n_words;
        pm1[0] -= 1;
        if (hostbig::cmp(twoq, pm1) == 0) {
            VMN_TRY(dev_zero(ctx, ctx->flags, sizeof(uint32_t)));
            int rc = VMN_ERR_ARG;
#define X(S_, NW_, LPE_)                                                                                                   \
    if constexpr (LPE_ == 1) {                                                                                             \
        if (m.S == S_) rc = launch_light(ctx, "member", k_jacobi_member<Cfg<S_, LPE_>>, grid_for(x->n), (const uint32_t*)x->d, x->n, (const uint32_t*)m.d_n, ctx->flags); \
    } else if constexpr (!Cfg<S_, LPE_>::WIDE) {                                                                           \
        if (m.S == S_) rc = launch_light(ctx, "member", k_jacobi_member_lanes<Cfg<S_, LPE_>>, egrid(m, x->n), (const uint32_t*)x->d, x->n, (const uint32_t*)m.d_n, ctx->flags); \
    }
            VMN_FOR_SIZES(X)
#undef X
            VMN_TRY(rc);
            uint32_t fl = 0;
            VMN_TRY(d2h(ctx, &fl, ctx->flags, sizeof(fl)));
            *all_members = fl ? 0 : 1;
            return VMN_OK;
        }
    }
    // x^q == 1 for every element: shared-exponent modpow, then compare with a broadcast of one
    DevTmp ew(ctx), pw(ctx);
    VMN_TRY(ew.alloc(m.NW * sizeof(uint32_t)));
    VMN_TRY(h2d(ctx, ew.p, g->Q.n_words.data(), m.NW * sizeof(uint32_t)));
    VMN_TRY(pw.alloc(2 * x->n * Wd * sizeof(uint32_t)));
    uint32_t* powers = pw.as<uint32_t>();
    uint32_t* ones = powers + x->n * Wd;
    VMN_TRY(modpow_words(ctx, m, x->d, ew.as<uint32_t>(), m.NW, 0, g->Q.nbits, x->n, powers));
    std::vector<uint32_t> idx(x->n, 0xffffffffu);
    VMN_TRY(gather_rows(ctx, m, powers, idx, m.d_one, ones));
    return compare_arrays(ctx, m, powers, ones, x->n, all_members);
}

extern "C" int vmn_group_mul_partials(vmn_group* grp, const uint8_t* partials_be, size_t k, uint8_t* out_be) {
    ARG_CHECK(grp && out_be && (partials_be || k == 0), "null argument");
    if (grp->curve) {
        vmn_garray* arr = nullptr;
        int ok = 1;
        VMN_TRY(vmn_garray_from_be(grp, partials_be, k, &arr, &ok));
        int rc = ok ? vmn_garray_prod(arr, out_be) : VMN_ERR_FORMAT;
        if (!ok) set_error("partial out of range");
        vmn_garray_free(arr);
        return rc;
    }
    const vmn_modulus& m = grp->P;
    const hostbig::Mont& hm = *m.hm;
    Big acc = hm.one;
    for (size_t i = 0; i < k; ++i) {
        Big v = hostbig::from_be(partials_be + i * grp->nbytes, grp->nbytes, m.NW);
        if (hostbig::cmp(v, m.n_words) >= 0) {
            set_error("partial %zu out of range", i);
            return VMN_ERR_FORMAT;
        }
        hm.mul(acc, acc, hm.to_mont(v));
    }
    hostbig::to_be(hm.from_mont(acc), out_be, grp->nbytes);
    return VMN_OK;
}
