// vmnhip_internal.h — host-side objects behind the opaque handles of include/vmnhip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <mutex>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/vmnhip.h"
#include "hostbig.h"
#include "hostnum64.h"
#include "hostcurve.h"
#include "hosttrace.h"

struct vmn_ctx;
namespace vmn {

void set_error(const char* fmt, ...);

#define VMN_HIP(call)                                                                         \
    do {                                                                                      \
        hipError_t err__ = (call);                                                            \
        if (err__ != hipSuccess) {                                                            \
            vmn::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(err__), __FILE__, __LINE__); \
            return err__ == hipErrorOutOfMemory ? VMN_ERR_NOMEM : VMN_ERR_DEVICE;             \
        }                                                                                     \
    } while (0)

// Entry-point prologue: calls of all threads on one context are serialised (the reference's protocol thread and
// its one helper thread, ShufflerElGamalSession.java:839-859, may call concurrently on distinct arrays); the
// mutex is recursive because entry points compose (permute -> gather, mul_partials -> from_be / prod).
#define VMN_ENTER(c)                                                \
    VMN_TRACE(__func__);                                            \
    std::lock_guard<std::recursive_mutex> guard__((c)->mu);         \
    VMN_HIP(hipSetDevice((c)->device))
// The lane a call runs on: the helper lane when the calling thread has announced itself as the helper of this context.
vmn_ctx* lane_of(vmn_ctx* c);
#define LANE(c) vmn::lane_of(c)

#define VMN_TRY(expr)                 \
    do {                              \
        int rc__ = (expr);            \
        if (rc__ != VMN_OK) return rc__; \
    } while (0)

struct TimingRec {
    std::string family;
    hipEvent_t start, stop;
    double mads = 0;           // v_mad_u64_u32 multiply-adds the launch executes (products x 2 S^2 ...), noted by the launch site
    double canon = 0;          // the same products in SURVEY.md §8d's canonical 32-bit multiply-accumulates (M(s) = 2 s^2 + s per product)
};

}  // namespace vmn

struct vmn_ctx {
    // A context is one "lane": a stream with its own pool, scratch, verdict words and mutex.  The context a caller
    // creates is the main lane; the ONE helper thread of the reference (ShufflerElGamalSession.java:839-859: mul + permute
    // beside a verification; the export thread of CCPoSW.java:114-123) gets a second lane on the same device
    // (vmn_ctx_helper_begin), so that its calls neither wait for the protocol thread's mutex nor queue behind its kernels.
    vmn_ctx* helper = nullptr;            // main lane: the helper lane, created on first use
    vmn_ctx* parent = nullptr;            // helper lane: its main lane
    hipEvent_t order_event = nullptr;     // helper lane: the latest "mark" on the main stream; the helper stream is ordered behind it
    bool marked = false;                  // helper lane: a mark has been recorded
    std::mutex order_mu;                  // helper lane: guards order_event / marked (both threads touch them)
    std::recursive_mutex mu;              // serialises the entry points of this lane (pool, scratch, flags and the stream are shared)
    // Main lane: blocks of fixed-base tables that were dropped (a group closed, a table evicted or regrown), kept for the next
    // table instead of going back to the driver.  A hipMalloc of gigabytes that FOLLOWS a hipFree of gigabytes was measured
    // at 0.6-3 s on this machine (profiles/r04_alloc_latency.txt: 3.1 s for 34 GB; the bench's end-to-end leg met 1.1 s
    // when it opened its group right after the previous leg had closed its own) while a first hipMalloc of 17 GB takes
    // 0.3 ms -- so the blocks stay here, bounded by VMN_TABLE_ARENA_BYTES (default 48 GB).
    std::vector<std::pair<void*, size_t>> table_arena;
    size_t table_arena_bytes = 0;
    std::mutex table_arena_mu;
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    int num_cus = 0;
    size_t wide_max = 0;                  // main lane: launches over at most this many elements use the wide geometry (vmnhip.hip: geom())
    size_t wide8_max = 0;                 //   ... and over at most this many the widest one (eight lanes per 2048-bit element)
    // grow-only scratch (window tables, temporaries)
    void* scratch = nullptr;
    size_t scratch_bytes = 0;
    uint32_t* flags = nullptr;            // small device word array for verdicts / range flags
    void* stage = nullptr;                // pinned host buffer (vmn::STAGE_BYTES): small device-to-host copies land here first --
                                          //   16 us instead of 27 us per read-back (tools/micro/copy_latency.hip); allocated on first use
    hipEvent_t stage_read = nullptr;      // recorded behind the latest host-to-device copy out of `stage` (the host waits for it before writing there again)
    bool stage_read_pending = false;
    // Small uploads (seeds, scalars, single elements: up to UP_SLOT bytes) go through a ring of pinned slots, one event each:
    // the host waits only when all UP_SLOTS uploads are still in flight.  Through the one `stage` buffer every small upload
    // waited for the previous one -- which sits behind everything queued on the stream: a stream synchronisation per scalar,
    // 55-70 us of idle device each (profiles/r04_timeline_p256_n10000_after.txt: 28 of them in one commit phase).
    static constexpr size_t UP_SLOT = 16384;
    static constexpr unsigned UP_SLOTS = 32;
    void* up_ring = nullptr;
    hipEvent_t up_done[UP_SLOTS] = {};
    bool up_pending[UP_SLOTS] = {};
    unsigned up_next = 0;
    static constexpr int PENDING_SLOTS = 2;          // multi-exponentiations in flight on this lane (vmn_pending): a verifier's A, F and its k_E products
    void* stage_pending[PENDING_SLOTS] = {nullptr, nullptr};     // their pinned landing buffers
    size_t stage_pending_bytes[PENDING_SLOTS] = {0, 0};
    bool stage_pending_busy[PENDING_SLOTS] = {false, false};
    // stream-ordered caching allocator: freed device blocks are kept by size and handed out again
    // (all work of a context is on one stream, so reuse is ordered after the previous user)
    std::map<size_t, std::vector<void*>> pool;
    std::map<const void*, size_t> block_class;     // main lane: every pooled block of the context -> the size class it belongs to
    std::mutex block_mu;                           //   (both lanes look blocks up: an array may be freed on the other lane)
    size_t pool_bytes = 0;
    size_t live_bytes = 0;                // handed out by pool_alloc and not yet returned
    std::unordered_set<const void*> lds_attr_set;
    bool timing = false;
    std::vector<vmn::TimingRec> recs;
    std::map<std::string, std::pair<long, double>> timing_acc;
    std::map<std::string, double> work_acc;    // executed multiply-adds per kernel family (while timing is on)
    std::map<std::string, double> canon_acc;   // the same in canonical 32-bit multiply-accumulates
    double next_canon = 0;                     // the same work in canonical 32-bit multiply-accumulates
    double next_mads = 0;                      // set by note_work() just before a launch, consumed by it
};

// Elliptic-curve group data (ECqPGroup): field constants on the device, see ec_kernels.h.
struct vmn_curve {
    std::string name;
    int S = 0, NW = 0;             // field limbs / packed words
    vmn::hostbig::Big p_words, b_words, gx_words, gy_words;
    uint32_t* d_consts = nullptr;  // one allocation holding p | one | rr | b | mp | mp2 | pm2
    const uint32_t* d_p = nullptr;
    const uint32_t* d_one = nullptr;
    const uint32_t* d_rr = nullptr;
    const uint32_t* d_b = nullptr;
    const uint32_t* d_mp = nullptr;
    const uint32_t* d_mp2 = nullptr;
    const uint32_t* d_pm2 = nullptr;
    const uint32_t* d_pp14 = nullptr;
    uint32_t n0inv = 0;
    vmn::num64::Mod* f64 = nullptr;     // the field on 64-bit limbs and the curve over it on the host: the sequential tail (Horner over
    vmn::num64::HostCurve host;         //   the windows) of a multi-exponentiation, 20 x faster there than on one GPU lane
    vmn::num64::Num rd_inv;             // 2^(-28 S) mod p: out of the device's Montgomery form, on the host (ec_export_few_host)
    uint32_t p1p = 0;              // limb 1 of the prime + 1 (ec_kernels.h mont_row)
    int ts_s = 0, ts_ewords = 0;   // Tonelli-Shanks (p = 1 mod 4): p - 1 = 2^ts_s Q; words of (Q - 1) / 2
    const uint32_t* d_ts_e = nullptr;
    const uint32_t* d_ts_c = nullptr;
};

// Device-resident constants of one odd modulus in M28 form.
struct vmn_modulus {
    int S = 0;                 // 28-bit limbs
    int NW = 0;                // 32-bit words of the packed form
    int LPE = 1;               // lanes per element (2 for 3072-bit moduli)
    int rows = 0;              // reduction rows of a product: R = 2^(28 rows); = S except in a wide geometry
    vmn_modulus* wide = nullptr;     // the wide geometry of the same rows (2048-bit moduli: 4 lanes per element), a view:
                                     // it shares every pointer of this object and owns nothing
    vmn_modulus* wide8 = nullptr;    // the same with eight lanes per element (2048-bit moduli)
    int W = 0;                 // words per element row in device memory
    const vmn_curve* ec = nullptr;   // non-null: the "elements" are curve points (rows of 3 field elements)
    int nbits = 0;
    uint32_t n0inv = 0;        // -N^{-1} mod 2^28
    uint32_t* d_n = nullptr;   // N as one device row (W words: per-lane shares, zero padded)
    uint32_t* d_rr = nullptr;  // R^2 mod N as a row, R = 2^(28 S)
    uint32_t* d_one = nullptr; // R mod N as a row == Montgomery form of 1
    vmn::hostbig::Big n_words; // NW words
    vmn::hostbig::Mont* hm = nullptr;     // host Montgomery context (32-bit words, R = 2^(32 NW))
    vmn::num64::Mod* hm64 = nullptr;      // the same on 64-bit limbs (the sequential tails: Horner of a multi-exponentiation)
};

struct vmn_group {
    vmn_ctx* ctx = nullptr;
    vmn_curve* curve = nullptr;   // EC group: P describes point rows (P.ec), Q the scalar field Z_n
    size_t nbytes = 0;         // host / wire width of a group element (ModPGroup) or of one coordinate (curves)
    size_t xbytes = 0;         // host / wire width of an exponent (ring element)
    vmn_modulus P;             // arithmetic mod p (group elements)
    vmn_modulus Q;             // arithmetic mod q (exponents)
    vmn::hostbig::Big g_words;
    // fixed-base tables, keyed by the base's big-endian bytes
    struct FixedTable {
        uint32_t* d_tab = nullptr;
        int wbits = 0;
        int nwin = 0;
        size_t bytes = 0;
        uint64_t last_use = 0;
        int uses = 0;              // calls served (a base that keeps coming back earns a larger window)
        hipEvent_t ready = nullptr;        // recorded behind the build; a user on another stream waits for it
        hipStream_t built_on = nullptr;
    };
    std::map<std::string, FixedTable> fixed;
    std::recursive_mutex tab_mu;          // the table cache is shared by the lanes of the context
    // least-recently-used tables are dropped beyond fixed_cache_limit() bytes (64 GB, env VMN_FIXED_CACHE_BYTES) (every proof brings a new base h_0)
    size_t fixed_bytes = 0;
    uint64_t fixed_clock = 0;
};

struct vmn_garray {
    vmn_group* grp = nullptr;
    uint32_t* d = nullptr;     // n * W words, M28 form mod p
    size_t n = 0;
    size_t bytes = 0;          // allocation size (pool key)
    vmn_ctx* lane = nullptr;   // the lane (stream + pool) the block was allocated on: it returns there, whoever frees it
};

struct vmn_rarray {
    vmn_group* grp = nullptr;
    uint32_t* d = nullptr;     // n * W words, M28 form mod q
    size_t n = 0;
    size_t bytes = 0;
    vmn_ctx* lane = nullptr;   // see vmn_garray
};
