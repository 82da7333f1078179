// vmnproofs.cpp — host-side drivers of include/vmnproofs.h: PoSBasicTW, PoSCBasicTW, CCPoSBasicW and the
// shuffler's arithmetic lines, written against the array-level C ABI (include/vmnhip.h) only.  Plain C++ (g++),
// no HIP here: every per-element operation is a vmn_* call that runs on the GPU; what stays on the host are the
// O(1) scalars of a proof, as in the reference (VCR scalar classes).
//
// Ordering note.  Calls of one context are stream-ordered and a call that returns a scalar blocks the host until
// everything queued before it has run.  Each method therefore (1) draws its randomness in the reference's
// order, (2) issues the calls that return scalars, (3) queues the long element-wise work, (4) does the host
// exponentiations of single elements while the GPU is busy, (5) returns without waiting (messages own arrays
// that may still be in flight; any later call is ordered behind them).  The VALUES are those of the reference's
// statement order; only independent statements were moved.
#include "../../include/vmnproofs.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <future>
#include <memory>
#include <string>
#include <system_error>
#include <vector>

#include "hostcurve.h"
#include "hostnum64.h"
#include "hosttrace.h"

using vmn::num64::Bytes;
using vmn::num64::Mod;
using vmn::num64::Num;

using vmn::num64::HostCurve;

namespace vmnp {

int fail(int code, const char* fmt, ...) {
    char buf[400];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    vmn_report_error(buf);
    return code;
}

#define TRY(expr)                           \
    do {                                    \
        int rc__ = (expr);                  \
        if (rc__ != VMN_OK) return rc__;    \
    } while (0)
#define REQUIRE(cond, msg)                                                   \
    do {                                                                     \
        if (!(cond)) return fail(VMN_ERR_ARG, "%s: %s", __func__, msg);      \
    } while (0)

// ---- owning handles --------------------------------------------------------------------------------------
struct GA {
    vmn_garray* p = nullptr;
    GA() {}
    GA(const GA&) = delete;
    GA& operator=(const GA&) = delete;
    GA(GA&& o) noexcept : p(o.p) { o.p = nullptr; }
    GA& operator=(GA&& o) noexcept {
        if (this != &o) {
            reset();
            p = o.p;
            o.p = nullptr;
        }
        return *this;
    }
    ~GA() { reset(); }
    void reset() {
        if (p) vmn_garray_free(p);
        p = nullptr;
    }
    vmn_garray** out() {
        reset();
        return &p;
    }
    vmn_garray* release() {
        vmn_garray* r = p;
        p = nullptr;
        return r;
    }
    operator const vmn_garray*() const { return p; }
};
struct RA {
    vmn_rarray* p = nullptr;
    RA() {}
    RA(const RA&) = delete;
    RA& operator=(const RA&) = delete;
    RA(RA&& o) noexcept : p(o.p) { o.p = nullptr; }
    ~RA() { reset(); }
    void reset() {
        if (p) vmn_rarray_free(p);
        p = nullptr;
    }
    vmn_rarray** out() {
        reset();
        return &p;
    }
    vmn_rarray* release() {
        vmn_rarray* r = p;
        p = nullptr;
        return r;
    }
    operator const vmn_rarray*() const { return p; }
};

std::vector<uint32_t> inverse_permutation(const uint32_t* pi, size_t n) {
    std::vector<uint32_t> inv(n);
    for (size_t i = 0; i < n; ++i) inv[pi[i]] = (uint32_t)i;
    return inv;
}
bool is_permutation(const uint32_t* pi, size_t n) {
    std::vector<uint8_t> seen(n, 0);
    for (size_t i = 0; i < n; ++i) {
        if (pi[i] >= n || seen[pi[i]]) return false;
        seen[pi[i]] = 1;
    }
    return true;
}

// ---- the group as the host sees it: sizes, Z_q scalars, single group elements ---------------------------------
// The exponentiations of SINGLE elements of a proof (g^alpha, pk^(-phi), A^v A' ...: 16 in a proof of a shuffle, a full-length
// one ~2 ms on one host core) are independent of each other within a phase: each runs on a worker thread of its own,
// started as soon as its operands exist -- beside the GPU calls of the phase, which block the calling thread -- and the
// phase joins them right before it uses the values.  At N = 10^6 this hides 30 ms of 1 s; at N = 10^4 it is a third of
// the proof.  Declare a HostJobs AFTER the objects its jobs write: it joins in its destructor, so an early return
// (TRY) never leaves a job writing into a dead object.  The jobs only call const members of HostGroup.
struct HostJobs {
    std::vector<std::future<int>> pending;
    template <class F>
    void start(F&& f) {
        if (pending.size() >= 32) {                 // (a very wide ciphertext: bounded number of threads, the rest inline)
            int rc = f();
            pending.emplace_back(std::async(std::launch::deferred, [rc] { return rc; }));
            return;
        }
        try {
            pending.emplace_back(std::async(std::launch::async, f));
        } catch (const std::system_error&) {        // no thread to be had: the job runs here and now
            int rc = f();
            pending.emplace_back(std::async(std::launch::deferred, [rc] { return rc; }));
        }
    }
    int join() {
        VMN_TRACE("host:jobs_join");
        int rc = VMN_OK;
        for (auto& f : pending) {
            int r = f.get();
            if (rc == VMN_OK) rc = r;
        }
        pending.clear();
        return rc;
    }
    ~HostJobs() { (void)join(); }
};

// An independent chain of ARRAY calls of a phase on the context's second lane (vmn_ctx_helper_*: its own stream, pool and
// lock), beside the calls the phase makes on the protocol thread's lane.  Below ~10^5 elements no kernel of a proof fills the
// device and a phase is the sum of the latencies of its dependent chains (at N = 10^4 a verifier's paired power of check (B)
// is one 7 ms launch on a third of the compute units while its multi-exponentiations wait behind it): two chains that do not
// depend on each other then run side by side.  start(): everything the calling thread has queued so far is what the job
// may rely on (vmn_ctx_helper_mark); join(): the job's stream has drained, its results can be used on any lane.
// Declare a LaneJob AFTER the objects its job writes (it joins in its destructor).  `enabled` false: the job runs inline.
struct LaneJob {
    std::future<int> fut;
    template <class F>
    int start(vmn_group* grp, bool enabled, F&& f) {
        vmn_ctx* ctx = enabled ? vmn_group_ctx(grp) : nullptr;
        if (!ctx || vmn_ctx_helper_mark(ctx) != VMN_OK) return f();
        try {
            fut = std::async(std::launch::async, [ctx, f]() -> int {
                const bool on_lane = vmn_ctx_helper_begin(ctx) == VMN_OK;      // (if not: the job still runs, on the caller's lane)
                const int rc = f();
                const int rc2 = on_lane ? vmn_ctx_helper_end(ctx) : (int)VMN_OK;
                return rc != VMN_OK ? rc : rc2;
            });
        } catch (const std::system_error&) {        // no thread to be had: the job runs here and now, on this lane
            return f();
        }
        return VMN_OK;
    }
    int join() {
        if (!fut.valid()) return VMN_OK;
        VMN_TRACE("lane:join");
        return fut.get();
    }
    ~LaneJob() { (void)join(); }
};
// arrays of at most this many elements leave the device idle enough for a second lane to pay (VMN_LANE_OVERLAP_MAX; 0 = never)
inline size_t lane_overlap_max() {
    static const size_t v = [] {
        const char* e = getenv("VMN_LANE_OVERLAP_MAX");
        return e && *e ? (size_t)strtoull(e, nullptr, 10) : (size_t)262144;
    }();
    return v;
}

struct HostGroup {
    vmn_group* grp = nullptr;
    bool ec = false;
    size_t eb = 0, xb = 0;            // element / exponent bytes
    size_t cw = 0;                    // bytes of the modulus (ModPGroup: = eb) or of one coordinate (curves: eb / 2)
    size_t ql = 0, pl = 0;            // 64-bit limbs of q / of the modulus resp. field prime
    Mod Zq, Zp;                       // Z_q scalars; Z_p = the modulus (ModPGroup) or the coordinate field (curves)
    HostCurve curve;
    int qbits = 0;
    bool safe_prime = false;          // ModPGroup with p = 2q + 1: membership of a single element = Jacobi symbol 1
    Bytes g;

    int init(vmn_group* g_) {
        grp = g_;
        ec = vmn_group_kind(grp) == 1;
        eb = vmn_group_elem_bytes(grp);
        xb = vmn_group_exp_bytes(grp);
        cw = ec ? eb / 2 : eb;
        ql = (xb + 7) / 8;
        pl = (cw + 7) / 8;
        Bytes qb(xb), pb(cw);
        TRY(vmn_group_get_order(grp, qb.data()));
        TRY(vmn_group_get_modulus(grp, pb.data()));
        Zq = Mod(vmn::num64::from_be(qb.data(), xb, ql));
        qbits = vmn::num64::bit_length(Zq.n);
        Zp = Mod(vmn::num64::from_be(pb.data(), cw, pl));
        if (!ec) {                                  // p = 2q + 1 ?
            Num t = Zq.n;
            t.resize(pl, 0);
            Num d = t;
            vmn::num64::add_in(t, d);
            Num o(pl, 0);
            o[0] = 1;
            vmn::num64::add_in(t, o);
            safe_prime = vmn::num64::cmp(t, Zp.n) == 0;
        }
        curve.F = &Zp;                 // (HostGroup objects are not copied after init)
        curve.cb = cw;
        curve.fl = pl;
        g.resize(eb);
        TRY(vmn_group_get_generator(grp, g.data()));
        return VMN_OK;
    }
    // ---- ring scalars
    Num ring_from(const uint8_t* be) const { return vmn::num64::from_be(be, xb, ql); }
    Bytes ring_bytes(const Num& a) const { return vmn::num64::to_bytes(a, xb); }
    Num reduce(const uint8_t* be, size_t n) const { return Zq.reduce(be, n); }
    Num mul_add(const Num& a, const Num& v, const Num& b) const { return Zq.add(Zq.mul(a, v), b); }   // a v + b

    // ---- single group elements (big-endian bytes, canonical => equality is memcmp)
    Bytes one() const {
        Bytes o(eb, ec ? 0xff : 0);               // EC: the point at infinity is all 0xff on the wire
        if (!ec) o[eb - 1] = 1;
        return o;
    }
    int el_exp(const Bytes& base, const uint8_t* e_be, size_t ebytes, Bytes& out) const {
        VMN_TRACE("host:el_exp");
        if (ec) {
            out = curve.exp(base, e_be, ebytes);
            return VMN_OK;
        }
        Num b = vmn::num64::from_be(base.data(), eb, pl);
        out = vmn::num64::to_bytes(Zp.pow(b, e_be, ebytes), eb);
        return VMN_OK;
    }
    int el_exp(const Bytes& base, const Num& e, Bytes& out) const {
        Bytes eb_ = ring_bytes(e);
        return el_exp(base, eb_.data(), eb_.size(), out);
    }
    int el_mul(const Bytes& a, const Bytes& b, Bytes& out) const {
        if (ec) {
            out = curve.add(a, b);
            return VMN_OK;
        }
        out = vmn::num64::to_bytes(Zp.mul(vmn::num64::from_be(a.data(), eb, pl), vmn::num64::from_be(b.data(), eb, pl)), eb);
        return VMN_OK;
    }
    int el_inv(const Bytes& a, Bytes& out) const {
        if (ec) {
            out = curve.negate(a);
            return VMN_OK;
        }
        out = vmn::num64::to_bytes(Zp.inv(vmn::num64::from_be(a.data(), eb, pl)), eb);
        return VMN_OK;
    }
    // Single elements that arrive from outside (a commitment's A', C', ...) are validated by the GPU import:
    // range / on-curve.  *ok = 0 when one of them is not a group element encoding.
    int check_elements(const std::vector<const Bytes*>& els, int* ok) const {
        Bytes flat;
        for (const Bytes* e : els) flat.insert(flat.end(), e->begin(), e->end());
        GA x;
        *ok = 1;
        TRY(vmn_garray_from_be(grp, flat.data(), els.size(), x.out(), ok));
        if (!*ok || ec) return VMN_OK;                 // curves: on the curve = in the group (cofactor 1)
        // ModPGroup: in range is not yet in the subgroup of order q (a handful of elements, on the host): the Jacobi
        // symbol when p = 2q + 1 (~0.1 ms each); x^q = 1 otherwise, every power on a thread of its own
        if (safe_prime) {
            for (const Bytes* e : els) {
                if (Zp.jacobi(vmn::num64::from_be(e->data(), eb, pl)) != 1) *ok = 0;
            }
            return VMN_OK;
        }
        const Bytes qb = vmn::num64::to_bytes(Zq.n, xb);
        Num one(pl, 0);
        one[0] = 1;
        std::vector<int> member(els.size(), 0);
        {
            HostJobs jobs;
            for (size_t k = 0; k < els.size(); ++k) {
                jobs.start([this, k, &els, &qb, &one, &member] {
                    member[k] = vmn::num64::cmp(Zp.pow(vmn::num64::from_be(els[k]->data(), eb, pl), qb.data(), qb.size()), one) == 0;
                    return (int)VMN_OK;
                });
            }
        }
        for (int mbr : member) if (!mbr) *ok = 0;
        return VMN_OK;
    }
    int el_div(const Bytes& a, const Bytes& b, Bytes& out) const {
        Bytes bi;
        TRY(el_inv(b, bi));
        return el_mul(a, bi, out);
    }
    // a^v * b  (a.expMul(v, b), PoSBasicTW.java:1016-1021)
    int el_expmul(const Bytes& a, const Bytes& v_be, const Bytes& b, Bytes& out) const {
        Bytes t;
        TRY(el_exp(a, v_be.data(), v_be.size(), t));
        return el_mul(t, b, out);
    }
};


}  // namespace vmnp
using namespace vmnp;

// ---- messages ------------------------------------------------------------------------------------------------
struct vmn_msg {
    struct Item {
        int kind = 0;
        vmn_garray* ga = nullptr;
        vmn_rarray* ra = nullptr;
        Bytes bytes;
        size_t count = 0, width = 0;
    };
    std::vector<Item> items;
    bool ec = false;                   // group elements are curve points: an element is node(leaf(x), leaf(y)) on the wire
    uint64_t serial = next_serial();   // distinguishes a message from an earlier one that lived at the same address
    static uint64_t next_serial() {
        static std::atomic<uint64_t> counter{1};
        return counter.fetch_add(1);
    }
    vmn_msg() {}
    explicit vmn_msg(bool ec_) : ec(ec_) {}
    ~vmn_msg() {
        for (auto& it : items) {
            if (it.ga) vmn_garray_free(it.ga);
            if (it.ra) vmn_rarray_free(it.ra);
        }
    }
    void push(GA& a) {
        Item it;
        it.kind = VMN_ITEM_GARRAY;
        it.ga = a.release();
        items.push_back(std::move(it));
    }
    void push(RA& a) {
        Item it;
        it.kind = VMN_ITEM_RARRAY;
        it.ra = a.release();
        items.push_back(std::move(it));
    }
    void push_bytes(int kind, const std::vector<Bytes>& els) {
        Item it;
        it.kind = kind;
        it.count = els.size();
        it.width = els.empty() ? 0 : els[0].size();
        for (auto& e : els) it.bytes.insert(it.bytes.end(), e.begin(), e.end());
        items.push_back(std::move(it));
    }
    void push_element(const Bytes& e) { push_bytes(VMN_ITEM_ELEMENTS, {e}); }
    void push_ring(const Bytes& e) { push_bytes(VMN_ITEM_RING, {e}); }
};

namespace vmnp {

const vmn_msg::Item* item_of(const vmn_msg* m, size_t i, int kind) {
    if (!m || i >= m->items.size() || m->items[i].kind != kind) return nullptr;
    return &m->items[i];
}
std::vector<Bytes> split(const vmn_msg::Item& it) {
    std::vector<Bytes> out;
    for (size_t k = 0; k < it.count; ++k) out.emplace_back(it.bytes.begin() + k * it.width, it.bytes.begin() + (k + 1) * it.width);
    return out;
}

// An explicitly supplied batching vector must consist of ebitlen-bit integers: A / F (expProd over ebitlen bits) and
// D (the full product of e) would otherwise be computed for different vectors.  The reference only ever derives e
// itself (setBatchVector(byte[] prgSeed), PoSBasicTW.java:533-538), so a wider entry is a caller's mistake.
int import_batch_vector(vmn_group* grp, const uint8_t* e_be, size_t n, int ebitlen, RA& e) {
    if (!e_be && n) return fail(VMN_ERR_ARG, "null batching vector");
    int ok = 1, bits = 0;
    TRY(vmn_rarray_from_be(grp, e_be, n, e.out(), &ok));
    if (!ok) return fail(VMN_ERR_FORMAT, "batching vector entry >= q");
    TRY(vmn_rarray_max_bits(e, &bits));
    if (bits > ebitlen) {
        e.reset();
        return fail(VMN_ERR_FORMAT, "batching vector entry of %d bits, wider than ebitlen = %d", bits, ebitlen);
    }
    return VMN_OK;
}
// Bit length to use for an exponent array that arrived in a message (a reply's k_E): every bit counts -- the reference
// parses full field elements (pField.toElementArray, PoSBasicTW.java:985-989) and h.expProd(k_E), B_shift.exp(k_E) use
// them whole (:1021, :1032) -- so the bound is measured, not assumed: an honest reply keeps its n_e + n_v + n_r + 1-bit
// exponentiations, anything longer is still computed exactly, and every check of one verification sees the same bits.
int received_bits(const vmn_rarray* a, int* bits) {
    int b = 0;
    TRY(vmn_rarray_max_bits(a, &b));
    *bits = b < 1 ? 1 : b;
    return VMN_OK;
}

// ---- N-sized random draws: host rows, or 32 bytes expanded on the device (vmn_random_source, vmnproofs.h) ----------
int random_ring_array(vmn_group* grp, const vmn_random_source& rs, size_t n, int qbits, int rbitlen, RA& out) {
    if (rs.array_seed && n > 1) {
        uint8_t seed[32];
        if (rs.array_seed(rs.user, seed) != 0) return fail(VMN_ERR_ARG, "random source failed");
        return vmn_rarray_from_prg(grp, seed, 32, n, qbits + rbitlen, out.out());
    }
    const uint8_t* rows = nullptr;
    if (!rs.ring_elements || rs.ring_elements(rs.user, n, &rows) != 0 || (!rows && n)) return fail(VMN_ERR_ARG, "random source failed");
    int ok = 1;
    TRY(vmn_rarray_from_be(grp, rows, n, out.out(), &ok));
    return ok ? VMN_OK : fail(VMN_ERR_FORMAT, "random source returned a value >= q");
}
int random_integer_array(vmn_group* grp, const vmn_random_source& rs, size_t n, int bits, RA& out) {
    if (rs.array_seed && n > 1) {
        uint8_t seed[32];
        if (rs.array_seed(rs.user, seed) != 0) return fail(VMN_ERR_ARG, "random source failed");
        return vmn_rarray_from_prg(grp, seed, 32, n, bits, out.out());
    }
    const uint8_t* rows = nullptr;
    if (!rs.integers || rs.integers(rs.user, n, bits, &rows) != 0 || (!rows && n)) return fail(VMN_ERR_ARG, "random source failed");
    int ok = 1;
    TRY(vmn_rarray_from_be(grp, rows, n, out.out(), &ok));
    return ok ? VMN_OK : fail(VMN_ERR_FORMAT, "random source returned a value >= q");
}

// An N-sized draw of a SHARDED prover.  With a seed source (vmn_random_source.array_seed) the draw IS its 32-byte seed:
// a rank generates the positions it holds and the rows it reads through the permutation (vmn_rarray_from_prg_range /
// _gather: the stream is counter mode), never all N values.  With host rows the whole array is kept, as before.
struct Draw {
    bool seeded = false;
    uint8_t seed[32] = {0};
    int bits = 0;
    RA all;
    // the values idx[0 .. n-1] of the draw
    int rows(vmn_group* grp, const uint32_t* idx, size_t n, RA& out) const {
        if (seeded) return vmn_rarray_from_prg_gather(grp, seed, 32, idx, n, bits, out.out());
        return vmn_rarray_gather(all, idx, n, out.out());
    }
    int range(vmn_group* grp, size_t lo, size_t hi, RA& out) const {
        if (seeded) return vmn_rarray_from_prg_range(grp, seed, 32, lo, hi - lo, bits, out.out());
        return vmn_rarray_copy_range(all, lo, hi, out.out());
    }
};
// ring elements (ring = true: bits(q) + rbitlen random bits reduced mod q) or integers of `bits` bits, n of them
int make_draw(vmn_group* grp, const vmn_random_source& rs, size_t n, bool ring, int bits, Draw& d) {
    d.bits = bits;
    if (rs.array_seed && n > 1) {
        if (rs.array_seed(rs.user, d.seed) != 0) return fail(VMN_ERR_ARG, "random source failed");
        d.seeded = true;
        return VMN_OK;
    }
    d.seeded = false;
    const uint8_t* rows = nullptr;
    const bool got = ring ? (rs.ring_elements && rs.ring_elements(rs.user, n, &rows) == 0)
                          : (rs.integers && rs.integers(rs.user, n, bits, &rows) == 0);
    if (!got || (!rows && n)) return fail(VMN_ERR_ARG, "random source failed");
    int ok = 1;
    TRY(vmn_rarray_from_be(grp, rows, n, d.all.out(), &ok));
    return ok ? VMN_OK : fail(VMN_ERR_FORMAT, "random source returned a value >= q");
}

// u_i = h_{idx(i)} g^{rsel_i}: the rows of a permutation commitment (PermutationCommitment.java:200-215) for n positions
int permutation_commitment_rows(vmn_group* grp, const uint8_t* g_be, const vmn_garray* h_full, const vmn_rarray* rsel,
                                const uint32_t* idx, size_t n, vmn_garray** u_out) {
    GA hsel, gr;
    TRY(vmn_group_exp_fixed(grp, g_be, rsel, gr.out()));
    TRY(vmn_garray_gather(h_full, idx, n, hsel.out()));
    return vmn_garray_mul(hsel, gr, u_out);
}
// w'_i = w_{idx(i)} pk^{ssel_i} for n positions and the 2 width components (ShufflerElGamalSession.java:273-278, 407)
int reencrypt_rows(vmn_group* grp, const uint8_t* pkey_be, size_t width, const vmn_garray* const* w_full, const vmn_rarray* const* ssel,
                   const uint32_t* idx, size_t n, vmn_garray** wp_out) {
    const size_t eb = vmn_group_elem_bytes(grp);
    std::vector<GA> res(2 * width);
    for (size_t c = 0; c < 2 * width; ++c) {
        GA factors, wsel;
        TRY(vmn_group_exp_fixed(grp, pkey_be + c * eb, ssel[c % width], factors.out()));
        TRY(vmn_garray_gather(w_full[c], idx, n, wsel.out()));
        TRY(vmn_garray_mul(wsel, factors, res[c].out()));
    }
    for (size_t c = 0; c < 2 * width; ++c) wp_out[c] = res[c].release();
    return VMN_OK;
}

// ---- what the three proofs share -----------------------------------------------------------------------------
// Sharding (include/vmnproofs.h vmn_comm): with a communicator set, every rank runs the SAME call sequence
// with random sources that return the SAME values; position-indexed arrays live as contiguous shards [lo, hi) of the N
// positions.  The public inputs that are read through the permutation (h for u = permute(h g^r, pi), w for w') are held in
// full on every rank, so the permuted arrays are local gathers and no element ever crosses a link.  The batching vector
// and the N-sized random arrays are NOT: they are PRG streams in counter mode, and a rank generates exactly the values
// it reads -- its own positions [lo, hi) and the rows e_{pi^-1(i)}, r_{pi(i)} (struct Draw) -- so the work per rank is
// O(N / world) (arrays handed over as host rows are still kept whole).  What crosses: one fixed-size all-gather per
// group of scalars -- partial products of expProd / prod, partial sums of the inner products, the carries of the two
// scans, each shard's last B, verdict bits.
struct ProofBase {
    HostGroup G;
    int vbitlen, ebitlen, rbitlen;
    int e_bits, eps_bits;              // prover-side bounds of values this object made itself (e from the PRG, epsilon)
    bool has_rs = false;
    vmn_random_source rs{};
    size_t N = 0;                      // size of THIS rank's arrays (= Ntot without a communicator)
    Bytes v_be;                        // the challenge as handed in (exponent of single elements)
    Num v;                             // ... and reduced mod q
    // sharding
    bool sharded = false;
    vmn_comm comm{};
    size_t Ntot = 0, lo = 0, hi = 0;
    RA e_full_own;                     // the whole batching vector when it was handed over as host rows (sharded provers permute it)
    Bytes e_seed;                      // ... or its PRG seed: the permuted rows are generated, not gathered
    // Every setter that changes an input of a verification (instance, commitment, batching vector, A / F) bumps the epoch;
    // a cached reply side (Prepared, below) is only used by the verify() of the epoch it was prepared in -- a verdict is
    // never computed from mixed inputs.  (The challenge is not part of the reply side.)
    uint64_t epoch = 1;
    void touch() { ++epoch; }
    // The ring scalars of a reply must be field elements: the reference parses them with pRing.toElement, which rejects
    // a value >= q (PoSBasicTW.java:985-989), so k_A + q makes ITS verifier answer false -- and this one too, whether the
    // reply came through vmn_msg_from_bytetree (which refuses it) or was assembled with vmn_msg_push_ring.
    bool ring_items_in_range(std::initializer_list<const vmn_msg::Item*> its) const {
        for (const vmn_msg::Item* it : its)
            for (size_t k = 0; it && k < it->count; ++k)
                if (vmn::num64::cmp(G.ring_from(it->bytes.data() + k * it->width), G.Zq.n) >= 0) return false;
        return true;
    }

    int init(vmn_group* grp, int vb, int ebl, int rb, const vmn_random_source* r) {
        TRY(G.init(grp));
        vbitlen = vb;
        ebitlen = ebl;
        rbitlen = rb;
        e_bits = std::min(ebl, G.qbits);
        eps_bits = std::min(ebl + vb + rb, G.qbits);
        if (r) {
            if (!r->ring_elements || !r->integers) return fail(VMN_ERR_ARG, "random source lacks a callback");
            rs = *r;
            has_rs = true;
        }
        return VMN_OK;
    }
    int set_comm(const vmn_comm* c) {
        if (!c || !c->all_gather || c->world < 1 || c->rank < 0 || c->rank >= c->world) return fail(VMN_ERR_ARG, "bad communicator");
        if (N) return fail(VMN_ERR_ARG, "the communicator must be set before the instance");
        comm = *c;
        sharded = true;
        return VMN_OK;
    }
    // the N positions of the whole proof and this rank's share of them
    void set_total(size_t ntot) {
        Ntot = ntot;
        if (sharded) {
            vmn_shard_bounds(ntot, comm.world, comm.rank, &lo, &hi);
        } else {
            lo = 0;
            hi = ntot;
        }
        N = hi - lo;
    }
    // An array argument of a sharded proof is either this rank's shard (N elements) or the whole array (Ntot elements,
    // replicated), of which the shard is cut here.  `own` keeps a cut alive.
    int local_garray(const vmn_garray* a, GA& own, const vmn_garray** out, const char* what) {
        const size_t n = vmn_garray_size(a);
        if (n == N && (!sharded || N == Ntot)) {
            *out = a;
            return VMN_OK;
        }
        if (sharded && n == Ntot) {
            TRY(vmn_garray_copy_range(a, lo, hi, own.out()));
            *out = own;
            return VMN_OK;
        }
        if (n == N) {
            *out = a;
            return VMN_OK;
        }
        return fail(VMN_ERR_ARG, "%s: %zu elements, expected this rank's %zu or all %zu", what, n, N, Ntot);
    }
    int local_rarray(const vmn_rarray* a, RA& own, const vmn_rarray** out, const char* what) {
        const size_t n = vmn_rarray_size(a);
        if (sharded && n == Ntot && N != Ntot) {
            TRY(vmn_rarray_copy_range(a, lo, hi, own.out()));
            *out = own;
            return VMN_OK;
        }
        if (n == N) {
            *out = a;
            return VMN_OK;
        }
        return fail(VMN_ERR_ARG, "%s: %zu elements, expected this rank's %zu or all %zu", what, n, N, Ntot);
    }
    // Two kinds of overlap.  (1) HOST work beside device work -- the squaring chain of a new fixed-base table: always worth it
    // for small arrays (prepare_base_table).  (2) two chains of KERNELS side by side: measured at N = 10^4 it LOSES
    // (profiles/r04_lane_overlap.txt: a lone wave64 already keeps its SIMD's 16-lane ALU busy 4 cycles of every ~5.5, so a
    // "latency-bound" chain is VALU-bound per SIMD, the wide geometries put a wave on every SIMD of the chip, and the wave that
    // arrives second gets the leftover issue slots: a bucket level beside the powers of check (B) took 5.3 ms instead of
    // 0.9 ms and the phase 16.2 ms instead of 11.7 ms).  It stays available for experiments: VMN_LANE_OVERLAP_COMPUTE=1.
    bool prefetch_tables() const { return N > 0 && N <= lane_overlap_max(); }
    // Curves are the exception: one point per lane, so an array of N <= 32 768 points is N / 64 waves on a chip of 1024 SIMDs
    // and two chains of such kernels find SIMDs of their own (the two scalar multiplications of check (B) at N = 10^4:
    // 1.58 + 1.55 ms one after the other, profiles/r04_timeline_p256_n10000.txt).  VMN_LANE_OVERLAP_EC_MAX moves that bound.
    bool overlap_lanes() const {
        static const bool on = [] {
            const char* e = getenv("VMN_LANE_OVERLAP_COMPUTE");
            return e && *e == '1';
        }();
        static const size_t ec_max = [] {
            const char* e = getenv("VMN_LANE_OVERLAP_EC_MAX");
            return e && *e ? (size_t)strtoull(e, nullptr, 10) : (size_t)32768;
        }();
        if (N == 0) return false;
        if (G.ec) return N <= ec_max;
        return on && N <= lane_overlap_max();
    }
    // the table of a per-proof base (h_0), built on the second lane as soon as the base is known: its squaring chain is
    // milliseconds of sequential host work that would otherwise sit in front of the first use (commit's bridging commitments)
    LaneJob h0_table_job;
    Bytes per_proof_base;                                    // h_0 of a prover: its table goes back to the arena with this object
    void release_base_table() {
        if (per_proof_base.empty()) return;
        (void)h0_table_job.join();
        (void)vmn_group_release_fixed(G.grp, per_proof_base.data());
        per_proof_base.clear();
    }
    int prepare_base_table(const Bytes& base) {
        per_proof_base = base;
        if (!prefetch_tables()) return VMN_OK;              // (large arrays hide the chain; curves: a one-lane doubling chain on the device, 2.3 ms)
        (void)h0_table_job.join();
        const Bytes b = base;                                // (a copy: the job may outlive the derived object's fields)
        vmn_group* grp = G.grp;
        const size_t n = N;
        return h0_table_job.start(grp, true, [grp, b, n] { return vmn_group_precompute_fixed(grp, b.data(), n, 2); });
    }
    int need_rs() const { return has_rs ? VMN_OK : fail(VMN_ERR_ARG, "this proof object was created without a random source (verifier)"); }
    // N-sized draws: `out` is this rank's shard; `full` (may be null) remembers the draw for reads through the permutation.
    int draw_ring_array(RA& out, Draw* full = nullptr) {
        VMN_TRACE("proof:draw_ring_array");
        TRY(need_rs());
        if (!sharded) return random_ring_array(G.grp, rs, Ntot, G.qbits, rbitlen, out);
        Draw local;
        Draw& d = full ? *full : local;
        TRY(make_draw(G.grp, rs, Ntot, true, G.qbits + rbitlen, d));
        return d.range(G.grp, lo, hi, out);
    }
    int draw_integers(int bits, RA& out) {
        VMN_TRACE("proof:draw_integers");
        TRY(need_rs());
        if (!sharded) return random_integer_array(G.grp, rs, Ntot, bits, out);
        Draw d;
        TRY(make_draw(G.grp, rs, Ntot, false, bits, d));
        return d.range(G.grp, lo, hi, out);
    }
    int draw_ring_element(Num& out) {
        TRY(need_rs());
        const uint8_t* rows = nullptr;
        if (rs.ring_elements(rs.user, 1, &rows) != 0 || !rows) return fail(VMN_ERR_ARG, "random source failed");
        out = G.ring_from(rows);
        if (vmn::num64::cmp(out, G.Zq.n) >= 0) return fail(VMN_ERR_FORMAT, "random source returned a value >= q");
        return VMN_OK;
    }
    int set_challenge(const uint8_t* vb, size_t n) {
        if (!vb || !n) return fail(VMN_ERR_ARG, "null challenge");
        v_be.assign(vb, vb + n);
        v = G.reduce(vb, n);
        return VMN_OK;
    }
    // the batching vector: this rank's shard in e; sharded objects keep the whole vector as well (e' = permute(e, pi^-1))
    int keep_batch_vector(RA& all, RA& e) {
        touch();
        if (!sharded) {
            e.reset();
            e.p = all.release();
            return VMN_OK;
        }
        TRY(vmn_rarray_copy_range(all, lo, hi, e.out()));
        e_full_own.reset();
        e_full_own.p = all.release();
        return VMN_OK;
    }
    int batch_vector_seed(const uint8_t* seed, size_t seedlen, RA& e) {
        if (!Ntot) return fail(VMN_ERR_ARG, "batching vector before the instance (size unknown)");
        if (sharded) {                                   // this rank's positions only; e' is generated row by row (below)
            touch();
            e_full_own.reset();
            e_seed.assign(seed, seed + seedlen);
            e.reset();
            return vmn_rarray_from_prg_range(G.grp, seed, seedlen, lo, N, ebitlen, e.out());
        }
        RA all;
        TRY(vmn_rarray_from_prg(G.grp, seed, seedlen, Ntot, ebitlen, all.out()));
        return keep_batch_vector(all, e);
    }
    int batch_vector(const uint8_t* e_be, RA& e) {
        RA all;
        TRY(import_batch_vector(G.grp, e_be, Ntot, ebitlen, all));
        e_seed.clear();
        return keep_batch_vector(all, e);
    }
    // e' = permute(e, pi^-1), this rank's positions
    int permuted_batch_vector(const RA& e, const std::vector<uint32_t>& piinv, RA& ipe) {
        VMN_TRACE("proof:permuted_batch_vector");
        if (!sharded) return vmn_rarray_permute(e, piinv.data(), ipe.out());
        if (!e_seed.empty()) return vmn_rarray_from_prg_gather(G.grp, e_seed.data(), e_seed.size(), piinv.data() + lo, N, ebitlen, ipe.out());
        return vmn_rarray_gather(e_full_own, piinv.data() + lo, N, ipe.out());
    }
    // g^a for a ring scalar; through the cached fixed-base table when the group is a curve (one launch), on the
    // host for ModPGroup
    int gexp(const Bytes& base, const Num& e, Bytes& out) const { return G.el_exp(base, e, out); }

    template <typename P>
    int mark_malformed(P& prep, const vmn_msg* rep) const {
        prep.malformed = true;
        prep.rep = rep;
        prep.serial = rep->serial;
        prep.epoch = epoch;
        return VMN_OK;
    }
    // ---- exchanges between the ranks: every rank contributes `mine` (same length everywhere) -----------------------
    int exchange(const Bytes& mine, std::vector<Bytes>& all) {
        all.clear();
        if (!sharded) {
            all.push_back(mine);
            return VMN_OK;
        }
        Bytes recv((size_t)comm.world * mine.size());
        if (comm.all_gather(comm.user, mine.data(), mine.size(), recv.data()) != 0) return fail(VMN_ERR_DEVICE, "all-gather failed");
        for (int k = 0; k < comm.world; ++k) all.emplace_back(recv.begin() + (size_t)k * mine.size(), recv.begin() + (size_t)(k + 1) * mine.size());
        return VMN_OK;
    }
    bool rank_nonempty(int k) const {
        size_t a, b;
        vmn_shard_bounds(Ntot, sharded ? comm.world : 1, k, &a, &b);
        return b > a;
    }
    // One exchange for a phase: group elements to multiply over the ranks (partial products), ring scalars to add or to
    // multiply, values to collect per rank (scan carries, last elements), flags to AND.
    struct Round {
        ProofBase& pb;
        std::vector<Bytes*> mul_el;
        std::vector<Num*> sum_r, mul_r;
        std::vector<std::pair<Bytes, std::vector<Bytes>*>> per_rank;     // (mine, all ranks' values)
        std::vector<int*> flags;
        explicit Round(ProofBase& p) : pb(p) {}
        void product(Bytes& el) { mul_el.push_back(&el); }
        void products(std::vector<Bytes>& els) {
            for (auto& e : els) mul_el.push_back(&e);
        }
        void sum(Num& x) { sum_r.push_back(&x); }
        void ring_product(Num& x) { mul_r.push_back(&x); }
        void collect(const Bytes& mine, std::vector<Bytes>& all) { per_rank.emplace_back(mine, &all); }
        void all_true(int& flag) { flags.push_back(&flag); }
        int run() {
            const HostGroup& G = pb.G;
            Bytes mine;
            for (Bytes* e : mul_el) mine.insert(mine.end(), e->begin(), e->end());
            for (Num* x : sum_r) {
                Bytes b = G.ring_bytes(*x);
                mine.insert(mine.end(), b.begin(), b.end());
            }
            for (Num* x : mul_r) {
                Bytes b = G.ring_bytes(*x);
                mine.insert(mine.end(), b.begin(), b.end());
            }
            for (auto& pr : per_rank) mine.insert(mine.end(), pr.first.begin(), pr.first.end());
            for (int* f : flags) mine.push_back(*f ? 1 : 0);
            if (mine.empty()) return VMN_OK;
            std::vector<Bytes> all;
            TRY(pb.exchange(mine, all));
            size_t off = 0;
            for (Bytes* e : mul_el) {
                Bytes acc(all[0].begin() + off, all[0].begin() + off + G.eb);
                for (size_t k = 1; k < all.size(); ++k) {
                    Bytes t;
                    TRY(G.el_mul(acc, Bytes(all[k].begin() + off, all[k].begin() + off + G.eb), t));
                    acc.swap(t);
                }
                *e = acc;
                off += G.eb;
            }
            for (Num* x : sum_r) {
                Num acc = G.ring_from(all[0].data() + off);
                for (size_t k = 1; k < all.size(); ++k) acc = G.Zq.add(acc, G.ring_from(all[k].data() + off));
                *x = acc;
                off += G.xb;
            }
            for (Num* x : mul_r) {
                Num acc = G.ring_from(all[0].data() + off);
                for (size_t k = 1; k < all.size(); ++k) acc = G.Zq.mul(acc, G.ring_from(all[k].data() + off));
                *x = acc;
                off += G.xb;
            }
            for (auto& pr : per_rank) {
                pr.second->clear();
                for (auto& a : all) pr.second->emplace_back(a.begin() + off, a.begin() + off + pr.first.size());
                off += pr.first.size();
            }
            for (int* f : flags) {
                int ok = 1;
                for (auto& a : all) ok = ok && a[off];
                *f = ok;
                off += 1;
            }
            return VMN_OK;
        }
    };

    // x = b.recLin(e'), y = e'.prods() over ALL positions: local scans, one exchange of the carries (the map of a shard is
    // x -> x E + X with E = the product of its e', X = its local result), a multiply-add fix-up on the ranks above 0.
    // x_in / y_in: the values entering this shard (x_{lo-1}, y_{lo-1}; 0 and 1 on the first).
    int scans(const RA& b, const RA& ipe, RA& x, RA& y, Num& d, Bytes& x_in, Bytes& y_in) {
        VMN_TRACE("proof:scans");
        Bytes dloc(G.xb, 0);
        TRY(vmn_rarray_rec_lin(b, ipe, x.out(), dloc.data()));
        TRY(vmn_rarray_prods(ipe, y.out()));
        x_in.assign(G.xb, 0);
        y_in.assign(G.xb, 0);
        y_in[G.xb - 1] = 1;
        if (!sharded) {
            d = G.ring_from(dloc.data());
            return VMN_OK;
        }
        Bytes mine(2 * G.xb, 0);                                   // (E, X) of this shard; an empty shard is the identity map (1, 0)
        mine[G.xb - 1] = 1;
        if (N) {
            TRY(vmn_rarray_get(y, N - 1, mine.data()));
            memcpy(mine.data() + G.xb, dloc.data(), G.xb);
        }
        std::vector<Bytes> all;
        TRY(exchange(mine, all));
        Num xin(G.ql, 0), yin(G.ql, 0), dd(G.ql, 0);
        yin[0] = 1;
        for (int k = 0; k < comm.world; ++k) {
            const Num E = G.ring_from(all[k].data()), X = G.ring_from(all[k].data() + G.xb);
            if (k < comm.rank) {
                xin = G.mul_add(xin, E, X);
                yin = G.Zq.mul(yin, E);
            }
            dd = G.mul_add(dd, E, X);
        }
        d = dd;
        x_in = G.ring_bytes(xin);
        y_in = G.ring_bytes(yin);
        if (comm.rank > 0 && N) {                                  // x_i = x_i^loc + x_in * P_i,  y_i = y_in * P_i  (P = local prods)
            RA xf, yf;
            TRY(vmn_rarray_mul_add(y, x_in.data(), x, xf.out()));
            TRY(vmn_rarray_mul_add(y, y_in.data(), nullptr, yf.out()));
            x.reset();
            y.reset();
            x.p = xf.release();
            y.p = yf.release();
        }
        return VMN_OK;
    }

    // the commitments B, B' of PoS and PoSC (PoSBasicTW.java:583-648, PoSCBasicTW.java:400-470): x, y are consumed;
    // x_in / y_in are what shiftPush pushes in front: (0, 1), or the carries into this shard
    int bridging_commitments(const Bytes& g, const Bytes& h0, RA& x, RA& y, const Bytes& x_in, const Bytes& y_in, const RA& beta,
                             const RA& epsilon, GA& B, GA& Bp) {
        VMN_TRACE("proof:bridging_commitments");
        GA g_exp_x, h0_exp_y;
        TRY(vmn_group_exp_fixed(G.grp, g.data(), x, g_exp_x.out()));
        TRY(vmn_group_exp_fixed(G.grp, h0.data(), y, h0_exp_y.out()));
        TRY(vmn_garray_mul(g_exp_x, h0_exp_y, B.out()));
        g_exp_x.reset();
        h0_exp_y.reset();
        RA xp, yp, xp_mul_eps, beta_add_prod, yp_mul_eps;
        TRY(vmn_rarray_shift_push(x, x_in.data(), xp.out()));
        TRY(vmn_rarray_shift_push(y, y_in.data(), yp.out()));
        x.reset();
        y.reset();
        TRY(vmn_rarray_mul(xp, epsilon, xp_mul_eps.out()));
        TRY(vmn_rarray_add(beta, xp_mul_eps, beta_add_prod.out()));
        GA g_exp_beta_add_prod, h0_exp_yp_mul_eps;
        TRY(vmn_group_exp_fixed(G.grp, g.data(), beta_add_prod, g_exp_beta_add_prod.out()));
        TRY(vmn_rarray_mul(yp, epsilon, yp_mul_eps.out()));
        TRY(vmn_group_exp_fixed(G.grp, h0.data(), yp_mul_eps, h0_exp_yp_mul_eps.out()));
        TRY(vmn_garray_mul(g_exp_beta_add_prod, h0_exp_yp_mul_eps, Bp.out()));
        return VMN_OK;
    }
    // check (B): B_i^v B'_i == g^{k_B,i} B_{i-1}^{k_E,i}, B_{-1} = h0 (PoSBasicTW.java:1023-1042).  Queues the
    // element-wise work and returns the two sides; the comparison (which blocks) is left to the caller.  `prev` = the
    // element in front of this shard's B (h0, or the last B of the shard below).
    int bridging_sides(const Bytes& g, const Bytes& prev, const vmn_garray* B, const vmn_garray* Bp, const vmn_rarray* k_B,
                       const vmn_rarray* k_E, int kE_bits, GA& left, GA& right) {
        VMN_TRACE("proof:bridging_sides");
        TRY(bridging_left(B, Bp, left));
        return bridging_right(g, prev, B, k_B, k_E, kE_bits, right);
    }
    // Check (B) as ONE simultaneous power (modular groups): B_i^v B'_i = g^{k_B,i} B_{i-1}^{k_E,i}  <=>  (B_i^v (B_{i-1}^{-1})^{k_E,i}) B'_i = g^{k_B,i}
    // for invertible B_{i-1}; the squarings of the 256-bit and the 613-bit exponent are shared (vmn_garray_exp2) and the
    // inverses of the whole array cost a few products per element (vmn_garray_inv).  *done = 0 when some B is not invertible
    // (the residue 0 -- no group element): the caller then evaluates the two sides separately, as the reference does.
    // (arrays that do not fill the device twice over are latency-bound: the inversion's scans and its host round trip cost
    // more than the squarings they save, and the paired launch below does better -- profiles/r03_pair_sweep.txt;
    // VMN_COMBINED_MIN moves the threshold -- the tests run the combined form at their sizes with it)
    // Curves (round 4): the same with a negation for the inverse (free) and k_ec_mulvar2's one chain of doublings for the two
    // scalar multiplications -- from the size on where the second lane no longer hides one of them behind the other.
    bool combined_form_pays() const {
        const char* env = getenv("VMN_COMBINED_MIN");
        const size_t min_n = env ? (size_t)strtoull(env, nullptr, 10) : (G.ec ? (size_t)32769 : (size_t)131073);
        return N >= min_n;
    }
    int bridging_combined(const Bytes& g, const Bytes& prev, const vmn_garray* B, const vmn_garray* Bp, const vmn_rarray* k_B,
                          const vmn_rarray* k_E, int kE_bits, GA& left, GA& right, int* done) {
        *done = 0;
        if (!combined_form_pays()) return VMN_OK;
        Bytes prev_inv;
        TRY(G.el_inv(prev, prev_inv));
        bool prev_zero = true;
        for (uint8_t c : prev_inv) prev_zero = prev_zero && c == 0;
        if (prev_zero) return VMN_OK;
        GA Binv, Bs, L;
        if (vmn_garray_inv(B, Binv.out()) != VMN_OK) return VMN_OK;               // an element without inverse: the separate form
        TRY(vmn_garray_shift_push(Binv, prev_inv.data(), Bs.out()));
        Binv.reset();
        TRY(vmn_garray_exp2(B, v_be.data(), v_be.size(), Bs, k_E, kE_bits, L.out()));
        TRY(vmn_garray_mul(L, Bp, left.out()));
        TRY(vmn_group_exp_fixed(G.grp, g.data(), k_B, right.out()));
        *done = 1;
        return VMN_OK;
    }
    // Both sides at once for arrays that do not fill the device (modular groups, verify() called with the challenge set):
    // B^v and B_shift^(k_E) share one launch (vmn_garray_exp_pair) -- each alone leaves a third of the device idle at the
    // reference's demo size.  Up to 2 x 131 072 elements = every slot of the device once; VMN_PAIR_MAX moves the bound (0 = never).
    bool pair_form_pays() const {
        const char* env = getenv("VMN_PAIR_MAX");
        const size_t max_n = env ? (size_t)strtoull(env, nullptr, 10) : (size_t)131072;
        return !G.ec && N > 0 && N <= max_n;
    }
    int bridging_pair(const Bytes& g, const Bytes& prev, const vmn_garray* B, const vmn_garray* Bp, const vmn_rarray* k_B,
                      const vmn_rarray* k_E, int kE_bits, GA& left, GA& right) {
        GA g_exp_k_B, B_shift, B_exp_v, B_shift_exp_k_E;
        TRY(vmn_group_exp_fixed(G.grp, g.data(), k_B, g_exp_k_B.out()));
        TRY(vmn_garray_shift_push(B, prev.data(), B_shift.out()));
        TRY(vmn_garray_exp_pair(B, v_be.data(), v_be.size(), B_shift, k_E, kE_bits, B_exp_v.out(), B_shift_exp_k_E.out()));
        TRY(vmn_garray_mul(B_exp_v, Bp, left.out()));
        return vmn_garray_mul(g_exp_k_B, B_shift_exp_k_E, right.out());
    }
    // left side B_i^v B'_i: needs the challenge
    int bridging_left(const vmn_garray* B, const vmn_garray* Bp, GA& left) {
        GA B_exp_v;
        TRY(vmn_garray_exp_scalar(B, v_be.data(), v_be.size(), B_exp_v.out()));
        return vmn_garray_mul(B_exp_v, Bp, left.out());
    }
    // right side g^{k_B,i} B_{i-1}^{k_E,i}: needs the reply only, not the challenge
    int bridging_right(const Bytes& g, const Bytes& prev, const vmn_garray* B, const vmn_rarray* k_B, const vmn_rarray* k_E, int kE_bits,
                       GA& right) {
        GA g_exp_k_B, B_shift, B_shift_exp_k_E;
        TRY(vmn_group_exp_fixed(G.grp, g.data(), k_B, g_exp_k_B.out()));
        TRY(vmn_garray_shift_push(B, prev.data(), B_shift.out()));
        TRY(vmn_garray_exp_array(B_shift, k_E, kE_bits, B_shift_exp_k_E.out()));
        return vmn_garray_mul(g_exp_k_B, B_shift_exp_k_E, right.out());
    }
    // the last B of the whole proof and the element in front of this shard's B, from every rank's last local B
    void pick_B(const std::vector<Bytes>& lasts, const Bytes& h0, Bytes& Blast, Bytes& prev) const {
        prev = h0;
        Blast = h0;
        const int world = sharded ? comm.world : 1, me = sharded ? comm.rank : 0;
        for (int k = 0; k < world; ++k) {
            if (!rank_nonempty(k)) continue;
            if (k < me) prev = lasts[k];
            Blast = lasts[k];
        }
    }
    // this rank's last B (the unit when its shard is empty: never used then)
    int last_local(const vmn_garray* B, Bytes& out) const {
        out = G.one();
        if (N) TRY(vmn_garray_get(B, N - 1, out.data()));
        return VMN_OK;
    }
    // pk_c^{-k_c mod width} * t_c for the 2w components of a ciphertext-shaped value: the powers do not need t (started
    // early, one job each), the products do
    void pk_powers(HostJobs& jobs, const std::vector<Bytes>& pkey, const std::vector<Num>& k, std::vector<Bytes>& pw) const {
        const size_t width = pkey.size() / 2;
        pw.assign(pkey.size(), Bytes());
        for (size_t c = 0; c < pkey.size(); ++c) {
            const Bytes* base = &pkey[c];
            Bytes* out = &pw[c];
            const Num kc = G.Zq.neg(k[c % width]);
            jobs.start([this, base, out, kc] { return G.el_exp(*base, kc, *out); });
        }
    }
    int pk_finish(const std::vector<Bytes>& pw, const std::vector<Bytes>& t, std::vector<Bytes>& out) const {
        out.resize(pw.size());
        for (size_t c = 0; c < pw.size(); ++c) TRY(G.el_mul(pw[c], t[c], out[c]));
        return VMN_OK;
    }
    // this rank's partial products (the caller completes them over the ranks in its phase's Round)
    int expprod_multi(const std::vector<const vmn_garray*>& xs, const vmn_rarray* e, int bits, std::vector<Bytes>& out) const {
        Bytes flat(xs.size() * G.eb);
        TRY(vmn_garray_expprod_multi(xs.data(), xs.size(), e, bits, flat.data()));
        out.clear();
        for (size_t k = 0; k < xs.size(); ++k) out.emplace_back(flat.begin() + k * G.eb, flat.begin() + (k + 1) * G.eb);
        return VMN_OK;
    }
    // <xs[i], ys[i]> (or the sum of xs[i] where ys[i] is null) for several pairs, one round trip
    int inner_products(const std::vector<const vmn_rarray*>& xs, const std::vector<const vmn_rarray*>& ys, std::vector<Num>& out) const {
        Bytes flat(xs.size() * G.xb);
        TRY(vmn_rarray_inner_products(xs.data(), ys.data(), xs.size(), flat.data()));
        out.clear();
        for (size_t i = 0; i < xs.size(); ++i) out.push_back(G.ring_from(flat.data() + i * G.xb));
        return VMN_OK;
    }
    // The same in two halves (vmn_garray_expprod_multi_begin / vmn_pending_finish): between begin() and finish() the caller
    // queues the device work that does not need the products -- the host-side tail of a multi-exponentiation (one squaring per
    // exponent bit, ~1 ms at 2048 bits) then runs beside it instead of in front of it.
    struct PendingProds {
        vmn_pending* p = nullptr;
        size_t k = 0;
        PendingProds() {}
        PendingProds(const PendingProds&) = delete;
        PendingProds& operator=(const PendingProds&) = delete;
        ~PendingProds() {
            if (p) vmn_pending_free(p);
        }
        int begin(const std::vector<const vmn_garray*>& xs, const vmn_rarray* e, int bits) {
            k = xs.size();
            return vmn_garray_expprod_multi_begin(xs.data(), k, e, bits, &p);
        }
        int finish(const HostGroup& G, std::vector<Bytes>& out) {
            Bytes flat(k * G.eb);
            vmn_pending* q = p;
            p = nullptr;
            TRY(vmn_pending_finish(q, flat.data()));
            out.clear();
            for (size_t i = 0; i < k; ++i) out.emplace_back(flat.begin() + i * G.eb, flat.begin() + (i + 1) * G.eb);
            return VMN_OK;
        }
    };
    // the 2w components of a ciphertext array, each this rank's shard (cut from whole arrays where those were handed in)
    int local_components(const vmn_garray* const* arr, size_t k, std::vector<GA>& own, std::vector<const vmn_garray*>& out, const char* what) {
        if (!arr) return fail(VMN_ERR_ARG, "%s: null", what);
        own.clear();
        own.resize(k);
        out.assign(k, nullptr);
        for (size_t c = 0; c < k; ++c) {
            if (!arr[c]) return fail(VMN_ERR_ARG, "%s: component %zu is null", what, c);
            TRY(local_garray(arr[c], own[c], &out[c], what));
        }
        return VMN_OK;
    }
    int local_columns(const vmn_rarray* const* arr, size_t k, std::vector<RA>& own, std::vector<const vmn_rarray*>& out, const char* what) {
        own.clear();
        own.resize(k);
        out.assign(k, nullptr);
        for (size_t c = 0; c < k; ++c) {
            if (!arr[c]) return fail(VMN_ERR_ARG, "%s: column %zu is null", what, c);
            TRY(local_rarray(arr[c], own[c], &out[c], what));
        }
        return VMN_OK;
    }
};

}  // namespace vmnp

// ================================================================================================================
// PoSBasicTW
// ================================================================================================================
struct vmn_pos : ProofBase {
    Bytes g, h0;
    const vmn_garray* h = nullptr;           // this rank's generators
    GA h_own;
    std::vector<uint32_t> pi, piinv;
    bool prover = false;
    RA r, epsilon, e, ipe, b, beta;
    GA u_own;
    const vmn_garray* u = nullptr;
    Num alpha, gamma, delta, d;
    std::vector<Num> phi;
    Bytes Ap;
    size_t width = 0;
    std::vector<Bytes> pkey;
    std::vector<const vmn_garray*> w, wp;
    std::vector<GA> w_own, wp_own;
    std::vector<const vmn_rarray*> s;
    std::vector<RA> s_own;
    // verifier
    Bytes A;
    std::vector<Bytes> F;
    const vmn_garray* cB = nullptr;
    const vmn_garray* cBp = nullptr;
    Bytes cAp, cCp, cDp;
    std::vector<Bytes> cFp;

    int precompute(const uint8_t* g_be, const vmn_garray* h_, const uint32_t* pi_) {
        touch();
        VMN_TRACE("pos:precompute");
        REQUIRE(g_be && h_, "null argument");
        REQUIRE(vmn_garray_size(h_) > 0, "empty generator array");
        prover = pi_ != nullptr;
        // h is always the whole array (a prover reads it through the permutation; every rank needs h_0 and the size)
        set_total(vmn_garray_size(h_));
        g.assign(g_be, g_be + G.eb);
        h0.assign(G.eb, 0);
        TRY(vmn_garray_get(h_, 0, h0.data()));                   // (with a communicator h is always the whole array)
        TRY(local_garray(h_, h_own, &h, "h"));
        if (!prover) return VMN_OK;                              // verifier :394-402
        REQUIRE(is_permutation(pi_, Ntot), "pi is not a permutation of [0, N)");
        pi.assign(pi_, pi_ + Ntot);
        piinv = inverse_permutation(pi_, Ntot);
        // :446-465  u_i = g^{r_pi(i)} h_pi(i)
        Draw r_draw;
        TRY(draw_ring_array(r, sharded ? &r_draw : nullptr));
        TRY(draw_ring_element(alpha));
        TRY(draw_integers(ebitlen + vbitlen + rbitlen, epsilon));
        Bytes hp(G.eb), ga;
        HostJobs jobs;
        jobs.start([&] { return gexp(g, alpha, ga); });
        TRY(prepare_base_table(h0));                             // the table of h_0 (used by commit): built beside this phase
        // :481  A' = g^alpha prod h_i^eps_i -- its device part is queued first, the permutation commitment behind it: the
        // fixed-base powers of u then run while the host finishes the product
        PendingProds hp_pending;
        std::vector<Bytes> hp_out;
        auto make_u = [&]() -> int {
            if (!sharded) return vmn_permutation_commitment(G.grp, g.data(), h_, r, pi.data(), u_own.out());
            RA r_perm;                                           // r_{pi(i)}, i in [lo, hi)
            TRY(r_draw.rows(G.grp, pi.data() + lo, N, r_perm));
            return permutation_commitment_rows(G.grp, g.data(), h_, r_perm, pi.data() + lo, N, u_own.out());
        };
        TRY(hp_pending.begin({h}, epsilon, eps_bits));           // (the second lane is busy with the table of h_0: one stream, in order)
        TRY(make_u());
        u = u_own;
        TRY(hp_pending.finish(G, hp_out));
        hp = hp_out[0];
        Round rd(*this);
        rd.product(hp);
        TRY(rd.run());
        TRY(jobs.join());
        return G.el_mul(ga, hp, Ap);
    }
    int set_instance(const uint8_t* pkey_be, size_t width_, const vmn_garray* const* w_, const vmn_garray* const* wp_,
                     const vmn_rarray* const* s_) {
        touch();
        REQUIRE(pkey_be && width_ > 0 && Ntot > 0, "null argument or precompute not called");
        width = width_;
        TRY(local_components(w_, 2 * width, w_own, w, "w"));
        TRY(local_components(wp_, 2 * width, wp_own, wp, "w'"));
        pkey.clear();
        for (size_t c = 0; c < 2 * width; ++c) pkey.emplace_back(pkey_be + c * G.eb, pkey_be + (c + 1) * G.eb);
        s.clear();
        if (s_) TRY(local_columns(s_, width, s_own, s, "re-encryption exponents"));
        return VMN_OK;
    }
    // The part of commit() that does not depend on the batching vector: all its random draws (same order, same values)
    // and F' = pk^(-phi) prod (w'_i)^(eps_i).  A caller that derives the batching vector by hashing the instance
    // (PoSTW.java:118-130: ~1.5 GB of byte trees at N = 10^6) runs this while the hash is computed; commit() does it
    // itself otherwise.
    bool prepared = false;
    Bytes Cp_, Dp_;
    std::vector<Bytes> Fp_;
    struct CommitPrep {                    // between the two halves of commit_prepare (declared in this order: the jobs are joined first)
        std::vector<Bytes> pkpow;
        PendingProds prods;
        HostJobs jobs;
    };
    int commit_draws() {
        // randomness in the reference's order: b :583, beta :612, gamma :667, delta :673, phi :687
        TRY(draw_ring_array(b));
        TRY(draw_ring_array(beta));
        TRY(draw_ring_element(gamma));
        TRY(draw_ring_element(delta));
        phi.resize(width);
        for (auto& ph : phi) TRY(draw_ring_element(ph));
        return VMN_OK;
    }
    int commit_prepare_begin(CommitPrep& cp, bool draw = true) {
        REQUIRE(prover && width && !prepared, "commit_prepare needs a prover with the instance set, once per proof");
        if (draw) TRY(commit_draws());
        cp.jobs.start([&] { return gexp(g, gamma, Cp_); });                       // :667-679
        cp.jobs.start([&] { return gexp(g, delta, Dp_); });
        pk_powers(cp.jobs, pkey, phi, cp.pkpow);                                  // :687-690
        return cp.prods.begin(wp, epsilon, eps_bits);                             // :690 (the device part)
    }
    int commit_prepare_finish(CommitPrep& cp) {
        std::vector<Bytes> prods;
        TRY(cp.prods.finish(G, prods));
        Round rd(*this);
        rd.products(prods);
        TRY(rd.run());
        TRY(cp.jobs.join());
        TRY(pk_finish(cp.pkpow, prods, Fp_));
        prepared = true;
        return VMN_OK;
    }
    int commit_prepare() {
        VMN_TRACE("pos:commit_prepare");
        CommitPrep cp;
        TRY(commit_prepare_begin(cp));
        return commit_prepare_finish(cp);
    }
    int commit(vmn_msg** out) {
        VMN_TRACE("pos:commit");
        REQUIRE(out && prover && e.p && width, "commit needs a prover with instance and batching vector set");
        // not prepared ahead: F' is begun here and finished behind the bridging commitments, whose fixed-base powers keep the
        // device busy while the host completes the product
        CommitPrep cp;
        const bool prepare_here = !prepared;
        GA B, Bp;
        auto bridge = [&]() -> int {
            TRY(permuted_batch_vector(e, piinv, ipe));                            // :552-554
            RA x, y;
            Bytes x_in, y_in;
            TRY(scans(b, ipe, x, y, d, x_in, y_in));                              // :583-604
            TRY(h0_table_job.join());                                             // (the table of h_0, begun in precompute)
            return bridging_commitments(g, h0, x, y, x_in, y_in, beta, epsilon, B, Bp);   // :606-648 (queued)
        };
        if (prepare_here && overlap_lanes() && !sharded) {
            // small arrays: the scans and the bridging commitments (a chain of short launches and four fixed-base powers) on the
            // second lane, F' (a multi-exponentiation over w') on this one -- neither fills the device
            TRY(commit_draws());
            LaneJob bridge_job;
            TRY(bridge_job.start(G.grp, true, bridge));
            int rc = commit_prepare_begin(cp, false);
            if (rc == VMN_OK) rc = commit_prepare_finish(cp);
            TRY(bridge_job.join());
            TRY(rc);
        } else {
            if (prepare_here) TRY(commit_prepare_begin(cp));
            TRY(bridge());
            if (prepare_here) TRY(commit_prepare_finish(cp));
        }
        std::unique_ptr<vmn_msg> m(new vmn_msg(G.ec));
        m->push(B);
        m->push_element(Ap);
        m->push(Bp);
        m->push_element(Cp_);
        m->push_element(Dp_);
        m->push_bytes(VMN_ITEM_ELEMENTS, Fp_);
        *out = m.release();
        return VMN_OK;
    }
    int reply(const uint8_t* vb, size_t vbytes, vmn_msg** out) {
        VMN_TRACE("pos:reply");
        REQUIRE(out && prover && ipe.p && s.size() == width, "reply needs commit() and the re-encryption exponents");
        TRY(set_challenge(vb, vbytes));
        std::vector<const vmn_rarray*> xs{r, r}, ys{ipe, nullptr};                 // <r, e'>, sum r, and per column <s_c, e>
        for (size_t col = 0; col < width; ++col) {                               // (product-ring inner product: per column)
            xs.push_back(s[col]);
            ys.push_back(e);
        }
        std::vector<Num> scal;
        TRY(inner_products(xs, ys, scal));
        Num a = scal[0], c = scal[1];
        std::vector<Num> f(scal.begin() + 2, scal.end());
        Round rd(*this);
        rd.sum(a);
        rd.sum(c);
        for (auto& fc : f) rd.sum(fc);
        TRY(rd.run());
        std::vector<Bytes> kF;
        for (size_t col = 0; col < width; ++col) kF.push_back(G.ring_bytes(G.mul_add(f[col], v, phi[col])));
        Bytes vq = G.ring_bytes(v);
        RA k_B, k_E;
        TRY(vmn_rarray_mul_add(b, vq.data(), beta, k_B.out()));
        TRY(vmn_rarray_mul_add(ipe, vq.data(), epsilon, k_E.out()));
        std::unique_ptr<vmn_msg> m(new vmn_msg(G.ec));
        m->push_ring(G.ring_bytes(G.mul_add(a, v, alpha)));
        m->push(k_B);
        m->push_ring(G.ring_bytes(G.mul_add(c, v, gamma)));
        m->push_ring(G.ring_bytes(G.mul_add(d, v, delta)));
        m->push(k_E);
        m->push_bytes(VMN_ITEM_RING, kF);
        *out = m.release();
        return VMN_OK;
    }
    GA u_cut;
    int set_permutation_commitment(const vmn_garray* u_) {
        touch();
        REQUIRE(u_, "null argument");
        return local_garray(u_, u_cut, &u, "u");
    }
    int compute_af() {
        touch();
        VMN_TRACE("pos:compute_af");
        REQUIRE(u && e.p && width, "computeAF needs u, the instance and the batching vector");
        std::vector<const vmn_garray*> xs{u};
        xs.insert(xs.end(), w.begin(), w.end());
        A.clear();
        F.clear();
        (void)af_job.join();
        af_pending.reset(new PendingProds());
        if (overlap_lanes()) {                                                    // small arrays: on the second lane (see verify_prepare)
            PendingProds* pp = af_pending.get();
            const vmn_rarray* ee = e;
            const int bits = e_bits;
            return af_job.start(G.grp, true, [pp, xs, ee, bits]() -> int { return pp->begin(xs, ee, bits); });
        }
        return af_pending->begin(xs, e, e_bits);                                  // one sort of e for u and w (the device part)
    }
    // A and F are completed where they are first needed -- in verify_prepare, behind the reply side of check (B), whose powers
    // then run on the device while the host finishes these products (and the sharded form exchanges them)
    std::unique_ptr<PendingProds> af_pending;
    LaneJob af_job;                        // (declared after af_pending: joined before it is destroyed)
    bool af_begun() const { return af_pending || !A.empty(); }
    int finish_af() {
        TRY(af_job.join());
        if (!af_pending) return VMN_OK;
        std::unique_ptr<PendingProds> pend(std::move(af_pending));
        std::vector<Bytes> res;
        TRY(pend->finish(G, res));
        Round rd(*this);
        rd.products(res);
        TRY(rd.run());
        A = res[0];
        F.assign(res.begin() + 1, res.end());
        return VMN_OK;
    }
    int set_commitment(const vmn_msg* m) {
        touch();
        VMN_TRACE("pos:set_commitment");
        const vmn_msg::Item *iB = item_of(m, 0, VMN_ITEM_GARRAY), *iAp = item_of(m, 1, VMN_ITEM_ELEMENTS),
                            *iBp = item_of(m, 2, VMN_ITEM_GARRAY), *iCp = item_of(m, 3, VMN_ITEM_ELEMENTS),
                            *iDp = item_of(m, 4, VMN_ITEM_ELEMENTS), *iFp = item_of(m, 5, VMN_ITEM_ELEMENTS);
        REQUIRE(m && m->items.size() == 6 && iB && iAp && iBp && iCp && iDp && iFp, "commitment is not (B, A', B', C', D', F')");
        REQUIRE(vmn_garray_size(iB->ga) == N && vmn_garray_size(iBp->ga) == N, "B / B' not of size N");
        REQUIRE(iAp->count == 1 && iCp->count == 1 && iDp->count == 1 && iFp->count == 2 * width && iAp->width == G.eb &&
                    iFp->width == G.eb, "commitment scalars have the wrong shape");
        cB = iB->ga;
        cBp = iBp->ga;
        cAp = iAp->bytes;
        cCp = iCp->bytes;
        cDp = iDp->bytes;
        cFp = split(*iFp);
        std::vector<const Bytes*> els{&cAp, &cCp, &cDp};
        for (auto& f : cFp) els.push_back(&f);
        int ok = 1;
        TRY(G.check_elements(els, &ok));
        if (!ok) {
            cB = nullptr;
            return fail(VMN_ERR_FORMAT, "commitment holds a value that is not a group element");
        }
        return VMN_OK;
    }
    // ---- verification in two parts.  Everything that needs the REPLY but not the CHALLENGE -- the right side of check (B)
    // with its 613-bit per-element powers, the multi-exponentiations with k_E, the products over u, h, e, g^{k_A}, g^{k_C},
    // g^{k_D}, pk^{-k_F}: four fifths of the verifier's GPU time -- can run while the challenge is still being derived (the
    // verifier hashes 0.5 GB of commitment for it; the reply is on the bulletin board long before that is done).
    // vmn_pos_verify_prepare(reply) does that part and keeps the results; verify() does it itself when it was not called.
    struct Prepared {
        const vmn_msg* rep = nullptr;
        uint64_t serial = 0, epoch = 0;
        bool malformed = false;             // a ring scalar of the reply is >= q: the verdict is false, nothing else was computed
        Bytes gkA, gkC, gkD, C, D;
        std::vector<Bytes> kE_prods, pkpow;
        GA right, left;
        bool paired = false;               // verify() itself asked (the challenge is known): both sides of (B) were queued as a pair, `left` too
        bool deferred = false;             // the reply side of check (B) was left to verify() (which then takes the combined form, or the paired launch)
        Bytes prev;
        int kE_bits = 0;
        void clear() {
            rep = nullptr;
            serial = 0;
            malformed = false;
            right.reset();
            left.reset();
            paired = false;
            kE_prods.clear();
            pkpow.clear();
            deferred = false;
        }
    } prep;
    int verify_prepare(const vmn_msg* rep, bool defer_bridge = false) {
        VMN_TRACE("pos:verify_prepare");
        REQUIRE(cB && af_begun(), "verify_prepare needs computeAF and setCommitment");
        const vmn_msg::Item *ikA = item_of(rep, 0, VMN_ITEM_RING), *ikB = item_of(rep, 1, VMN_ITEM_RARRAY),
                            *ikC = item_of(rep, 2, VMN_ITEM_RING), *ikD = item_of(rep, 3, VMN_ITEM_RING),
                            *ikE = item_of(rep, 4, VMN_ITEM_RARRAY), *ikF = item_of(rep, 5, VMN_ITEM_RING);
        REQUIRE(rep && rep->items.size() == 6 && ikA && ikB && ikC && ikD && ikE && ikF, "reply is not (k_A, k_B, k_C, k_D, k_E, k_F)");
        REQUIRE(vmn_rarray_size(ikB->ra) == N && vmn_rarray_size(ikE->ra) == N && ikF->count == width && ikA->width == G.xb,
                "reply items have the wrong shape");
        prep.clear();
        if (!ring_items_in_range({ikA, ikC, ikD, ikF})) return mark_malformed(prep, rep);
        Num k_A = G.ring_from(ikA->bytes.data()), k_C = G.ring_from(ikC->bytes.data()), k_D = G.ring_from(ikD->bytes.data());
        std::vector<Num> k_F;
        for (auto& bts : split(*ikF)) k_F.push_back(G.ring_from(bts.data()));
        Bytes uprod(G.eb), hprod(G.eb), mylast, eprod_b(G.xb), t_h0, Blast, prev;
        Num eprod;
        int kE_bits = 0;
        std::vector<const vmn_garray*> xs{h};
        xs.insert(xs.end(), wp.begin(), wp.end());
        PendingProds kE_pending;
        std::promise<int> kE_begun;                                               // set by the lane job once the device part is queued
        std::future<int> kE_begun_f = kE_begun.get_future();
        HostJobs jobs;                                                            // (after everything its jobs touch)
        LaneJob kE_job;
        jobs.start([&] { return gexp(g, k_A, prep.gkA); });                       // (A) :1016-1021
        jobs.start([&] { return gexp(g, k_C, prep.gkC); });                       // (C) :1045-1048
        jobs.start([&] { return gexp(g, k_D, prep.gkD); });                       // (D) :1051-1054
        pk_powers(jobs, pkey, k_F, prep.pkpow);                                   // (F) :1057-1063
        // The reply side of check (B) :1030-1033 -- unless verify() follows at once and takes the combined form.  It needs the
        // element in front of this shard's B: h0 when there is one shard, and then it is queued here, behind the device part
        // of the products and in front of their host part (which it hides); with several shards it waits for the exchange.
        // Small arrays: the powers of check (B) are ONE long launch that fills a third of the device -- the multi-exponentiations
        // of this phase (and computeAF's before it) run on the second lane, so that this lane is free for those powers at once.
        TRY(received_bits(ikE->ra, &kE_bits));
        prep.deferred = defer_bridge && combined_form_pays();
        prep.kE_bits = kE_bits;
        prep.paired = defer_bridge && !prep.deferred && pair_form_pays();         // small arrays, challenge known: B^v rides along
        bool pair_launch = prep.paired;                                           // (modular groups: both powers in one launch)
        auto queue_bridge = [&](const Bytes& front) -> int {
            if (pair_launch) return bridging_pair(g, front, cB, cBp, ikB->ra, ikE->ra, kE_bits, prep.left, prep.right);
            if (!prep.deferred) return bridging_right(g, front, cB, ikB->ra, ikE->ra, kE_bits, prep.right);
            return VMN_OK;
        };
        // ONE job on the second lane (a lane is one in-order stream): the multi-exponentiations with k_E, then -- curves with the
        // challenge known -- B^v B', beside the reply side of check (B) on this lane
        const bool prods_aside = overlap_lanes();
        const bool left_aside = defer_bridge && G.ec && prods_aside && !prep.paired && !prep.deferred;
        if (left_aside) prep.paired = true;                                       // (verify() takes prep.left)
        if (prods_aside) {
            TRY(kE_job.start(G.grp, true, [&, left_aside]() -> int {
                const int rc = kE_pending.begin(xs, ikE->ra, kE_bits);    // first: its host-side tail (the Horner chains) then runs beside the powers
                kE_begun.set_value(rc);
                if (rc != VMN_OK) return rc;
                return left_aside ? bridging_left(cB, cBp, prep.left) : (int)VMN_OK;
            }));
        }
        // scalars that come back from the GPU (each blocks on the stream) ...
        TRY(vmn_garray_prod(u, uprod.data()));                                    // :1013
        TRY(vmn_garray_prod(h, hprod.data()));
        TRY(last_local(cB, mylast));
        TRY(vmn_rarray_prod(e, eprod_b.data()));                                  // :1014
        eprod = G.ring_from(eprod_b.data());
        if (!prods_aside) TRY(kE_pending.begin(xs, ikE->ra, kE_bits));            // :1021, :1063 — one sort of k_E (the device part)
        if (!sharded) TRY(queue_bridge(h0));
        TRY(finish_af());
        if (prods_aside) TRY(kE_begun_f.get());                                   // (queued, not finished: finish() waits for its event)
        TRY(kE_pending.finish(G, prep.kE_prods));
        // ... completed over the ranks in ONE exchange ...
        std::vector<Bytes> lasts;
        Round rd(*this);
        rd.product(uprod);
        rd.product(hprod);
        rd.products(prep.kE_prods);
        rd.ring_product(eprod);
        rd.collect(mylast, lasts);
        TRY(rd.run());
        pick_B(lasts, h0, Blast, prev);
        TRY(G.el_div(uprod, hprod, prep.C));
        jobs.start([&] {
            TRY(G.el_exp(h0, eprod, t_h0));
            return G.el_div(Blast, t_h0, prep.D);
        });
        prep.prev = prev;
        if (sharded) TRY(queue_bridge(prev));
        TRY(jobs.join());
        TRY(kE_job.join());
        lastC = prep.C;
        lastD = prep.D;
        prep.rep = rep;
        prep.serial = rep->serial;
        prep.epoch = epoch;
        return VMN_OK;
    }
    // The verifier's intermediate values, as PoSBasicTW exposes them (getA :716, getF :761, getC :949, getD :958; printed by
    // `vmnv -t PoS.A,...`, MixNetElGamalVerifyFiatShamirSession.java:880-932).  A and F exist after computeAF (its products
    // are finished here if they are still pending), C and D after verify / verify_prepare.
    Bytes lastC, lastD;
    int get_A(uint8_t* out) {
        REQUIRE(out && af_begun(), "getA needs computeAF");
        TRY(finish_af());
        memcpy(out, A.data(), G.eb);
        return VMN_OK;
    }
    int get_F(uint8_t* out) {
        REQUIRE(out && af_begun(), "getF needs computeAF");
        TRY(finish_af());
        for (size_t c = 0; c < F.size(); ++c) memcpy(out + c * G.eb, F[c].data(), G.eb);
        return VMN_OK;
    }
    int get_CD(uint8_t* out, bool want_d) {
        const Bytes& x = want_d ? lastD : lastC;
        REQUIRE(out && x.size() == G.eb, "getC / getD need a verify (or verify_prepare) that got as far as the reply");
        memcpy(out, x.data(), G.eb);
        return VMN_OK;
    }
    int verify(const vmn_msg* rep, int* verdict, int* five) {
        VMN_TRACE("pos:verify");
        REQUIRE(verdict && cB && af_begun() && !v_be.empty(), "verify needs computeAF, setCommitment and setChallenge");
        if (!rep || prep.rep != rep || prep.serial != rep->serial || prep.epoch != epoch) TRY(verify_prepare(rep, true));
        // the reply side is used ONCE, whatever happens below: with `paired` set its left side depends on the challenge, so a
        // verify() that fails half way must not leave it behind for a later call with another challenge
        struct ClearPrep {
            Prepared& p;
            ~ClearPrep() { p.clear(); }
        } clear_prep{prep};
        TRY(finish_af());
        if (prep.malformed) {                                                     // a ring scalar >= q: not a reply (:985-989)
            prep.clear();
            if (five) five[0] = five[1] = five[2] = five[3] = five[4] = 0;
            *verdict = 0;
            return VMN_OK;
        }
        Bytes lhsA, lhsC, lhsD, rhs;
        std::vector<Bytes> lhsF(2 * width), rF;
        GA left;
        int vB = 0;
        {
            HostJobs jobs;                                                        // the challenge side: four short powers
            jobs.start([&] { return G.el_expmul(A, v_be, cAp, lhsA); });          // (A) :1016-1021
            jobs.start([&] { return G.el_expmul(prep.C, v_be, cCp, lhsC); });     // (C)
            jobs.start([&] { return G.el_expmul(prep.D, v_be, cDp, lhsD); });     // (D)
            for (size_t c = 0; c < 2 * width; ++c) jobs.start([this, c, &lhsF] { return G.el_expmul(F[c], v_be, cFp[c], lhsF[c]); });
            int combined = 0;
            if (prep.deferred) {                                                  // (B) :1023-1042, both sides now
                const vmn_msg::Item *ikB = item_of(rep, 1, VMN_ITEM_RARRAY), *ikE = item_of(rep, 4, VMN_ITEM_RARRAY);
                TRY(bridging_combined(g, prep.prev, cB, cBp, ikB->ra, ikE->ra, prep.kE_bits, left, prep.right, &combined));
                if (!combined) TRY(bridging_right(g, prep.prev, cB, ikB->ra, ikE->ra, prep.kE_bits, prep.right));
            }
            if (prep.paired) left = std::move(prep.left);
            else if (!combined) TRY(bridging_left(cB, cBp, left));                     // :1028-1029
            TRY(vmn_garray_equals(left, prep.right, &vB));
            TRY(jobs.join());
        }
        TRY(G.el_mul(prep.gkA, prep.kE_prods[0], rhs));
        const int vA = lhsA == rhs;
        const int vC = lhsC == prep.gkC;
        const int vD = lhsD == prep.gkD;
        std::vector<Bytes> prods(prep.kE_prods.begin() + 1, prep.kE_prods.end());
        TRY(pk_finish(prep.pkpow, prods, rF));
        int vF = 1;
        for (size_t c = 0; c < 2 * width; ++c) vF = vF && lhsF[c] == rF[c];
        prep.clear();                                                             // the reply side is used once
        Round rv(*this);
        rv.all_true(vB);
        if (sharded) TRY(rv.run());
        if (five) {
            five[0] = vA;
            five[1] = vB;
            five[2] = vC;
            five[3] = vD;
            five[4] = vF;
        }
        *verdict = vA && vB && vC && vD && vF;                                    // no short-circuit :1065
        return VMN_OK;
    }
};

// ================================================================================================================
// PoSCBasicTW
// ================================================================================================================
struct vmn_posc : ProofBase {
    Bytes g, h0;
    const vmn_garray* h = nullptr;
    const vmn_garray* u = nullptr;
    const vmn_rarray* r = nullptr;
    GA h_own, u_own;
    RA r_own;
    std::vector<uint32_t> piinv;
    RA epsilon, e, ipe, b, beta;
    Num alpha, gamma, delta, d;
    const vmn_garray* cB = nullptr;
    const vmn_garray* cBp = nullptr;
    Bytes cAp, cCp, cDp;

    // h: the whole array; u / r: whole arrays or this rank's shards (a prover's pi is always the whole permutation)
    int set_instance(const uint8_t* g_be, const vmn_garray* h_, const vmn_garray* u_, const vmn_rarray* r_, const uint32_t* pi_) {
        touch();
        REQUIRE(g_be && h_ && u_, "null argument");
        set_total(vmn_garray_size(h_));
        REQUIRE(Ntot > 0, "h / u empty");
        g.assign(g_be, g_be + G.eb);
        h0.assign(G.eb, 0);
        TRY(vmn_garray_get(h_, 0, h0.data()));
        TRY(local_garray(h_, h_own, &h, "h"));
        TRY(local_garray(u_, u_own, &u, "u"));
        r = nullptr;
        piinv.clear();
        if (pi_) {
            REQUIRE(r_, "prover needs the commitment exponents r");
            TRY(local_rarray(r_, r_own, &r, "r"));
            REQUIRE(is_permutation(pi_, Ntot), "pi is not a permutation of [0, N)");
            piinv = inverse_permutation(pi_, Ntot);
            TRY(prepare_base_table(h0));                         // a prover will raise h_0 to N exponents in commit()
        }
        return VMN_OK;
    }
    // the part of commit() that does not depend on the batching vector (see vmn_pos::commit_prepare)
    bool prepared = false;
    Bytes Ap_, Cp_, Dp_;
    struct CommitPrep {                    // between the two halves of commit_prepare (see vmn_pos; the jobs are joined first)
        Bytes ga;
        PendingProds prod;
        HostJobs jobs;                     // the three host exponentiations run beside the multi-exponentiation
    };
    int commit_draws() {
        // randomness in the reference's order: b, alpha, epsilon, beta, gamma, delta (PoSCBasicTW.java:400-500)
        TRY(draw_ring_array(b));
        TRY(draw_ring_element(alpha));
        TRY(draw_integers(ebitlen + vbitlen + rbitlen, epsilon));
        TRY(draw_ring_array(beta));
        TRY(draw_ring_element(gamma));
        return draw_ring_element(delta);
    }
    int commit_prepare_begin(CommitPrep& cp, bool draw = true) {
        REQUIRE(!piinv.empty() && !prepared, "commit_prepare needs a prover instance, once per proof");
        if (draw) TRY(commit_draws());
        cp.jobs.start([this, &cp] { return gexp(g, alpha, cp.ga); });
        cp.jobs.start([&] { return gexp(g, gamma, Cp_); });
        cp.jobs.start([&] { return gexp(g, delta, Dp_); });
        return cp.prod.begin({h}, epsilon, eps_bits);
    }
    int commit_prepare_finish(CommitPrep& cp) {
        std::vector<Bytes> hp;
        TRY(cp.prod.finish(G, hp));
        Round rd(*this);
        rd.product(hp[0]);
        TRY(rd.run());
        TRY(cp.jobs.join());
        TRY(G.el_mul(cp.ga, hp[0], Ap_));
        prepared = true;
        return VMN_OK;
    }
    int commit_prepare() {
        CommitPrep cp;
        TRY(commit_prepare_begin(cp));
        return commit_prepare_finish(cp);
    }
    int commit(vmn_msg** out) {
        REQUIRE(out && !piinv.empty() && e.p, "commit needs a prover instance and the batching vector");
        CommitPrep cp;                                                            // (see vmn_pos::commit)
        const bool prepare_here = !prepared;
        GA B, Bp;
        auto bridge = [&]() -> int {
            TRY(permuted_batch_vector(e, piinv, ipe));
            RA x, y;
            Bytes x_in, y_in;
            TRY(scans(b, ipe, x, y, d, x_in, y_in));
            TRY(h0_table_job.join());
            return bridging_commitments(g, h0, x, y, x_in, y_in, beta, epsilon, B, Bp);
        };
        if (prepare_here && overlap_lanes() && !sharded) {
            TRY(commit_draws());
            LaneJob bridge_job;
            TRY(bridge_job.start(G.grp, true, bridge));
            int rc = commit_prepare_begin(cp, false);
            if (rc == VMN_OK) rc = commit_prepare_finish(cp);
            TRY(bridge_job.join());
            TRY(rc);
        } else {
            if (prepare_here) TRY(commit_prepare_begin(cp));
            TRY(bridge());
            if (prepare_here) TRY(commit_prepare_finish(cp));
        }
        std::unique_ptr<vmn_msg> m(new vmn_msg(G.ec));
        m->push(B);
        m->push_element(Ap_);
        m->push(Bp);
        m->push_element(Cp_);
        m->push_element(Dp_);
        *out = m.release();
        return VMN_OK;
    }
    int reply(const uint8_t* vb, size_t vbytes, vmn_msg** out) {
        REQUIRE(out && ipe.p && r, "reply needs commit()");
        TRY(set_challenge(vb, vbytes));
        std::vector<Num> scal;
        TRY(inner_products({r, r}, {ipe, nullptr}, scal));
        Num a = scal[0], c = scal[1];
        Round rd(*this);
        rd.sum(a);
        rd.sum(c);
        TRY(rd.run());
        Bytes vq = G.ring_bytes(v);
        RA k_B, k_E;
        TRY(vmn_rarray_mul_add(b, vq.data(), beta, k_B.out()));
        TRY(vmn_rarray_mul_add(ipe, vq.data(), epsilon, k_E.out()));
        std::unique_ptr<vmn_msg> m(new vmn_msg(G.ec));
        m->push_ring(G.ring_bytes(G.mul_add(a, v, alpha)));
        m->push(k_B);
        m->push_ring(G.ring_bytes(G.mul_add(c, v, gamma)));
        m->push_ring(G.ring_bytes(G.mul_add(d, v, delta)));
        m->push(k_E);
        *out = m.release();
        return VMN_OK;
    }
    int set_commitment(const vmn_msg* m) {
        touch();
        const vmn_msg::Item *iB = item_of(m, 0, VMN_ITEM_GARRAY), *iAp = item_of(m, 1, VMN_ITEM_ELEMENTS),
                            *iBp = item_of(m, 2, VMN_ITEM_GARRAY), *iCp = item_of(m, 3, VMN_ITEM_ELEMENTS),
                            *iDp = item_of(m, 4, VMN_ITEM_ELEMENTS);
        REQUIRE(m && m->items.size() == 5 && iB && iAp && iBp && iCp && iDp, "commitment is not (B, A', B', C', D')");
        REQUIRE(vmn_garray_size(iB->ga) == N && vmn_garray_size(iBp->ga) == N && iAp->count == 1 && iCp->count == 1 &&
                    iDp->count == 1 && iAp->width == G.eb, "commitment items have the wrong shape");
        cB = iB->ga;
        cBp = iBp->ga;
        cAp = iAp->bytes;
        cCp = iCp->bytes;
        cDp = iDp->bytes;
        int ok = 1;
        TRY(G.check_elements({&cAp, &cCp, &cDp}, &ok));
        if (!ok) {
            cB = nullptr;
            return fail(VMN_ERR_FORMAT, "commitment holds a value that is not a group element");
        }
        return VMN_OK;
    }
    // verification in two parts, as in vmn_pos: verify_prepare(reply) = everything that needs no challenge
    struct Prepared {
        const vmn_msg* rep = nullptr;
        uint64_t serial = 0, epoch = 0;
        bool malformed = false;             // a ring scalar of the reply is >= q: the verdict is false, nothing else was computed
        Bytes gkA, gkC, gkD, A, C, D, hk;
        GA right, left;
        bool paired = false;               // verify() itself asked (the challenge is known): both sides of (B) were queued as a pair, `left` too
        bool deferred = false;             // see vmn_pos
        Bytes prev;
        int kE_bits = 0;
        void clear() {
            rep = nullptr;
            serial = 0;
            malformed = false;
            right.reset();
            left.reset();
            paired = false;
            deferred = false;
        }
    } prep;
    int verify_prepare(const vmn_msg* rep, bool defer_bridge = false) {
        REQUIRE(cB && e.p, "verify_prepare needs the batching vector and setCommitment");
        const vmn_msg::Item *ikA = item_of(rep, 0, VMN_ITEM_RING), *ikB = item_of(rep, 1, VMN_ITEM_RARRAY),
                            *ikC = item_of(rep, 2, VMN_ITEM_RING), *ikD = item_of(rep, 3, VMN_ITEM_RING),
                            *ikE = item_of(rep, 4, VMN_ITEM_RARRAY);
        REQUIRE(rep && rep->items.size() == 5 && ikA && ikB && ikC && ikD && ikE, "reply is not (k_A, k_B, k_C, k_D, k_E)");
        REQUIRE(vmn_rarray_size(ikB->ra) == N && vmn_rarray_size(ikE->ra) == N && ikA->width == G.xb, "reply items have the wrong shape");
        prep.clear();
        if (!ring_items_in_range({ikA, ikC, ikD})) return mark_malformed(prep, rep);
        Num k_A = G.ring_from(ikA->bytes.data()), k_C = G.ring_from(ikC->bytes.data()), k_D = G.ring_from(ikD->bytes.data());
        Bytes uprod(G.eb), hprod(G.eb), mylast, eprod_b(G.xb), Blast, prev, t_h0;
        prep.A.assign(G.eb, 0);
        prep.hk.assign(G.eb, 0);
        Num eprod;
        int kE_bits = 0;
        PendingProds a_pending, hk_pending;                                       // (see vmn_pos::verify_prepare)
        std::vector<Bytes> a_out, hk_out;
        std::promise<int> prods_begun;
        std::future<int> prods_begun_f = prods_begun.get_future();
        HostJobs jobs;                                                            // (after everything its jobs touch)
        LaneJob prods_job;
        jobs.start([&] { return gexp(g, k_A, prep.gkA); });                       // beside the GPU calls below
        jobs.start([&] { return gexp(g, k_C, prep.gkC); });
        jobs.start([&] { return gexp(g, k_D, prep.gkD); });
        TRY(received_bits(ikE->ra, &kE_bits));
        prep.deferred = defer_bridge && combined_form_pays();
        prep.kE_bits = kE_bits;
        prep.paired = defer_bridge && !prep.deferred && pair_form_pays();         // small arrays, challenge known: B^v rides along
        bool pair_launch = prep.paired;
        auto queue_bridge = [&](const Bytes& front) -> int {
            if (pair_launch) return bridging_pair(g, front, cB, cBp, ikB->ra, ikE->ra, kE_bits, prep.left, prep.right);
            if (!prep.deferred) return bridging_right(g, front, cB, ikB->ra, ikE->ra, kE_bits, prep.right);
            return VMN_OK;
        };
        const bool prods_aside = overlap_lanes();                                 // the multi-exponentiations on the second lane
        const bool left_aside = defer_bridge && G.ec && prods_aside && !prep.paired && !prep.deferred;
        if (left_aside) prep.paired = true;                                       // (see vmn_pos::verify_prepare)
        if (prods_aside) {
            TRY(prods_job.start(G.grp, true, [&, left_aside]() -> int {
                int rc = a_pending.begin({u}, e, e_bits);                         // :660
                if (rc == VMN_OK) rc = hk_pending.begin({h}, ikE->ra, kE_bits);
                prods_begun.set_value(rc);
                if (rc != VMN_OK) return rc;
                return left_aside ? bridging_left(cB, cBp, prep.left) : (int)VMN_OK;
            }));
        } else {
            TRY(vmn_garray_expprod(u, e, e_bits, prep.A.data()));                 // :660
        }
        TRY(vmn_garray_prod(u, uprod.data()));
        TRY(vmn_garray_prod(h, hprod.data()));
        TRY(last_local(cB, mylast));
        TRY(vmn_rarray_prod(e, eprod_b.data()));
        eprod = G.ring_from(eprod_b.data());
        if (!prods_aside) TRY(hk_pending.begin({h}, ikE->ra, kE_bits));
        if (!sharded) TRY(queue_bridge(h0));
        if (prods_aside) {
            TRY(prods_begun_f.get());
            TRY(a_pending.finish(G, a_out));
            prep.A = a_out[0];
        }
        TRY(hk_pending.finish(G, hk_out));
        prep.hk = hk_out[0];
        std::vector<Bytes> lasts;
        Round rd(*this);
        rd.product(prep.A);
        rd.product(uprod);
        rd.product(hprod);
        rd.product(prep.hk);
        rd.ring_product(eprod);
        rd.collect(mylast, lasts);
        TRY(rd.run());
        pick_B(lasts, h0, Blast, prev);
        TRY(G.el_div(uprod, hprod, prep.C));
        jobs.start([&] {
            TRY(G.el_exp(h0, eprod, t_h0));
            return G.el_div(Blast, t_h0, prep.D);
        });
        prep.prev = prev;
        if (sharded) TRY(queue_bridge(prev));
        TRY(jobs.join());
        TRY(prods_job.join());
        prep.rep = rep;
        prep.serial = rep->serial;
        prep.epoch = epoch;
        return VMN_OK;
    }
    // A, C, D of the verifier (private fields of PoSCBasicTW: A = u.expProd(e) :676, C :718-723, D :724-727), for tests and a
    // test-vector dump; available after verify / verify_prepare
    Bytes lastA, lastC, lastD;
    int get_ACD(uint8_t* out, int which) {
        const Bytes& x = which == 0 ? lastA : which == 1 ? lastC : lastD;
        REQUIRE(out && x.size() == G.eb, "getA / getC / getD need a verify (or verify_prepare) that got as far as the reply");
        memcpy(out, x.data(), G.eb);
        return VMN_OK;
    }
    int verify(const vmn_msg* rep, int* verdict) {
        REQUIRE(verdict && cB && e.p && !v_be.empty(), "verify needs the batching vector, setCommitment and setChallenge");
        *verdict = 0;
        if (!rep || prep.rep != rep || prep.serial != rep->serial || prep.epoch != epoch) TRY(verify_prepare(rep, true));
        struct ClearPrep {                     // used once, on every exit path (see vmn_pos::verify)
            Prepared& p;
            ~ClearPrep() { p.clear(); }
        } clear_prep{prep};
        lastA = prep.A;
        lastC = prep.C;
        lastD = prep.D;
        if (prep.malformed) {                                                     // a ring scalar >= q: not a reply
            prep.clear();
            *verdict = 0;
            return VMN_OK;
        }
        Bytes lhsA, lhsC, lhsD, rhs;
        int vB = 0;
        bool a_ok = false;
        {
            HostJobs jobs;
            jobs.start([&] { return G.el_expmul(prep.A, v_be, cAp, lhsA); });     // (A) :676-682
            jobs.start([&] { return G.el_expmul(prep.C, v_be, cCp, lhsC); });     // (C) :718-723
            jobs.start([&] { return G.el_expmul(prep.D, v_be, cDp, lhsD); });     // (D) :724-727
            GA left;
            int combined = 0;
            if (prep.deferred) {                                                  // (B) :685-715, both sides now
                const vmn_msg::Item *ikB = item_of(rep, 1, VMN_ITEM_RARRAY), *ikE = item_of(rep, 4, VMN_ITEM_RARRAY);
                TRY(bridging_combined(g, prep.prev, cB, cBp, ikB->ra, ikE->ra, prep.kE_bits, left, prep.right, &combined));
                if (!combined) TRY(bridging_right(g, prep.prev, cB, ikB->ra, ikE->ra, prep.kE_bits, prep.right));
            }
            if (prep.paired) left = std::move(prep.left);
            else if (!combined) TRY(bridging_left(cB, cBp, left));
            TRY(vmn_garray_equals(left, prep.right, &vB));
            TRY(jobs.join());
        }
        TRY(G.el_mul(prep.gkA, prep.hk, rhs));
        a_ok = lhsA == rhs;                                                       // short-circuit :682: nothing after (A) counts when it fails
        const int vC = lhsC == prep.gkC;
        const int vD = lhsD == prep.gkD;
        prep.clear();
        Round rv(*this);
        rv.all_true(vB);
        if (sharded) TRY(rv.run());
        *verdict = a_ok && vB && vC && vD;
        return VMN_OK;
    }
};

// ================================================================================================================
// CCPoSBasicW
// ================================================================================================================
struct vmn_ccpos : ProofBase {
    Bytes g;
    const vmn_garray* h = nullptr;
    const vmn_garray* u = nullptr;
    const vmn_rarray* r = nullptr;
    GA h_own, u_own;
    RA r_own;
    std::vector<uint32_t> piinv;
    size_t width = 0;
    std::vector<Bytes> pkey;
    std::vector<const vmn_garray*> w, wp;
    std::vector<GA> w_own, wp_own;
    std::vector<const vmn_rarray*> s;
    std::vector<RA> s_own;
    RA epsilon, e, ipe;
    Num alpha;
    std::vector<Num> beta;
    Bytes cAp;
    std::vector<Bytes> cBp;
    bool have_commitment = false;
    Bytes A;
    std::vector<Bytes> B, AB;
    bool raised = false, have_ab = false;

    int set_instance(const uint8_t* g_be, const vmn_garray* h_, const vmn_garray* u_, const uint8_t* pkey_be, size_t width_,
                     const vmn_garray* const* w_, const vmn_garray* const* wp_, const vmn_rarray* r_, const uint32_t* pi_,
                     const vmn_rarray* const* s_) {
        touch();
        REQUIRE(g_be && h_ && u_ && pkey_be && width_ > 0, "null argument");
        set_total(vmn_garray_size(h_));
        REQUIRE(Ntot > 0, "h / u empty");
        width = width_;
        g.assign(g_be, g_be + G.eb);
        TRY(local_garray(h_, h_own, &h, "h"));
        TRY(local_garray(u_, u_own, &u, "u"));
        TRY(local_components(w_, 2 * width, w_own, w, "w"));
        TRY(local_components(wp_, 2 * width, wp_own, wp, "w'"));
        pkey.clear();
        for (size_t c = 0; c < 2 * width; ++c) pkey.emplace_back(pkey_be + c * G.eb, pkey_be + (c + 1) * G.eb);
        piinv.clear();
        s.clear();
        r = nullptr;
        if (pi_) {
            REQUIRE(r_ && s_, "prover needs r and s");
            TRY(local_rarray(r_, r_own, &r, "r"));
            REQUIRE(is_permutation(pi_, Ntot), "pi is not a permutation of [0, N)");
            piinv = inverse_permutation(pi_, Ntot);
            TRY(local_columns(s_, width, s_own, s, "re-encryption exponents"));
        }
        return VMN_OK;
    }
    // the part of commit() that does not depend on the batching vector -- here all of its arithmetic: the two
    // multi-exponentiations with epsilon (see vmn_pos::commit_prepare)
    bool prepared = false;
    Bytes Ap_;
    std::vector<Bytes> Bp_;
    int commit_prepare() {
        VMN_TRACE("ccpos:commit_prepare");
        REQUIRE(!piinv.empty() && !prepared, "commit_prepare needs a prover instance, once per proof");
        TRY(draw_ring_element(alpha));                                            // :360-375
        TRY(draw_integers(ebitlen + vbitlen + rbitlen, epsilon));
        beta.resize(width);
        for (auto& bt : beta) TRY(draw_ring_element(bt));
        std::vector<const vmn_garray*> xs{h};
        xs.insert(xs.end(), wp.begin(), wp.end());
        std::vector<Bytes> eps_prods, pkpow;
        Bytes ga;
        HostJobs jobs;                                 // g^alpha and pk^(-beta) run beside the multi-exponentiation
        jobs.start([&] { return gexp(g, alpha, ga); });
        pk_powers(jobs, pkey, beta, pkpow);
        TRY(expprod_multi(xs, epsilon, eps_bits, eps_prods));                     // :377, :391 — one sort of epsilon
        Round rd(*this);
        rd.products(eps_prods);
        TRY(rd.run());
        TRY(jobs.join());
        TRY(G.el_mul(ga, eps_prods[0], Ap_));
        std::vector<Bytes> prods(eps_prods.begin() + 1, eps_prods.end());
        TRY(pk_finish(pkpow, prods, Bp_));
        prepared = true;
        return VMN_OK;
    }
    int commit(vmn_msg** out) {
        VMN_TRACE("ccpos:commit");
        REQUIRE(out && !piinv.empty() && e.p, "commit needs a prover instance and the batching vector");
        if (!prepared) TRY(commit_prepare());
        TRY(permuted_batch_vector(e, piinv, ipe));                                // :350
        std::unique_ptr<vmn_msg> m(new vmn_msg(G.ec));
        m->push_element(Ap_);
        m->push_bytes(VMN_ITEM_ELEMENTS, Bp_);
        *out = m.release();
        return VMN_OK;
    }
    int reply(const uint8_t* vb, size_t vbytes, vmn_msg** out) {
        VMN_TRACE("ccpos:reply");
        REQUIRE(out && ipe.p && r && s.size() == width, "reply needs commit()");
        TRY(set_challenge(vb, vbytes));
        std::vector<const vmn_rarray*> xs{r}, ys{ipe};
        for (size_t col = 0; col < width; ++col) {
            xs.push_back(s[col]);
            ys.push_back(e);
        }
        std::vector<Num> scal;
        TRY(inner_products(xs, ys, scal));
        Num a = scal[0];
        std::vector<Num> f(scal.begin() + 1, scal.end());
        Round rd(*this);
        rd.sum(a);
        for (auto& fc : f) rd.sum(fc);
        TRY(rd.run());
        std::vector<Bytes> kB;
        for (size_t col = 0; col < width; ++col) kB.push_back(G.ring_bytes(G.mul_add(f[col], v, beta[col])));
        Bytes vq = G.ring_bytes(v);
        RA k_E;
        TRY(vmn_rarray_mul_add(ipe, vq.data(), epsilon, k_E.out()));
        std::unique_ptr<vmn_msg> m(new vmn_msg(G.ec));
        m->push_ring(G.ring_bytes(G.mul_add(a, v, alpha)));
        m->push_bytes(VMN_ITEM_RING, kB);
        m->push(k_E);
        *out = m.release();
        return VMN_OK;
    }
    int set_commitment(const vmn_msg* m) {
        VMN_TRACE("ccpos:set_commitment");
        touch();
        const vmn_msg::Item *iAp = item_of(m, 0, VMN_ITEM_ELEMENTS), *iBp = item_of(m, 1, VMN_ITEM_ELEMENTS);
        REQUIRE(m && m->items.size() == 2 && iAp && iBp && iAp->count == 1 && iBp->count == 2 * width && iAp->width == G.eb &&
                    iBp->width == G.eb, "commitment is not (A', B')");
        cAp = iAp->bytes;
        cBp = split(*iBp);
        std::vector<const Bytes*> els{&cAp};
        for (auto& f : cBp) els.push_back(&f);
        int ok = 1;
        TRY(G.check_elements(els, &ok));
        if (!ok) return fail(VMN_ERR_FORMAT, "commitment holds a value that is not a group element");
        have_commitment = true;
        return VMN_OK;
    }
    GA ru_own;
    int compute_ab(const vmn_garray* raisedu) {
        VMN_TRACE("ccpos:compute_ab");
        touch();
        REQUIRE(u && e.p && width, "computeAB needs the instance and the batching vector");
        raised = raisedu != nullptr;
        // the products are begun here and finished in verify_prepare, behind the device part of the k_E products (see vmn_pos)
        ab_pending.reset(new PendingProds());
        if (!raised) {
            std::vector<const vmn_garray*> xs{u};
            xs.insert(xs.end(), w.begin(), w.end());
            TRY(ab_pending->begin(xs, e, e_bits));
        } else {
            const vmn_garray* ru = nullptr;
            TRY(local_garray(raisedu, ru_own, &ru, "raised commitment"));
            // w.mul(raisedu): the base-group array multiplies every component (:502); then one sort of e
            std::vector<GA> tmp(2 * width);
            std::vector<const vmn_garray*> xs;
            for (size_t c = 0; c < 2 * width; ++c) {
                TRY(vmn_garray_mul(w[c], ru, tmp[c].out()));
                xs.push_back(tmp[c]);
            }
            TRY(ab_pending->begin(xs, e, e_bits));         // (the products w u^rho are queued; their blocks return to the pool in stream order)
        }
        have_ab = true;
        return VMN_OK;
    }
    std::unique_ptr<PendingProds> ab_pending;
    int finish_ab() {
        if (!ab_pending) return VMN_OK;
        std::unique_ptr<PendingProds> pend(std::move(ab_pending));
        std::vector<Bytes> res;
        TRY(pend->finish(G, res));
        Round rd(*this);
        rd.products(res);
        TRY(rd.run());
        if (!raised) {
            A = res[0];
            B.assign(res.begin() + 1, res.end());
        } else {
            AB = res;
        }
        return VMN_OK;
    }
    // A and B of computeAB (:493-506): plain form A (1 element) then B (2 width); raised form AB (2 width elements)
    int get_AB(uint8_t* out, size_t* count) {
        REQUIRE(out && count && have_ab, "getAB needs computeAB");
        TRY(finish_ab());
        std::vector<const Bytes*> els;
        if (!raised) {
            els.push_back(&A);
            for (auto& b : B) els.push_back(&b);
        } else {
            for (auto& ab : AB) els.push_back(&ab);
        }
        for (size_t c = 0; c < els.size(); ++c) memcpy(out + c * G.eb, els[c]->data(), G.eb);
        *count = els.size();
        return VMN_OK;
    }
    // verification in two parts, as in vmn_pos: verify_prepare(reply, raisedh, rho) = everything that needs no challenge --
    // here ALL the GPU work of verify(): the multi-exponentiations with k_E
    struct Prepared {
        const vmn_msg* rep = nullptr;
        uint64_t serial = 0, epoch = 0;
        bool malformed = false;             // a ring scalar of the reply is >= q: the verdict is false, nothing else was computed
        Bytes gkA, Ap_rho, g_term;
        std::vector<Bytes> pkpow, kE_prods, prods;
        const vmn_garray* raisedh = nullptr;       // what the raised form was prepared with: verify() must be handed the same
        Bytes rho;
        void clear() {
            rep = nullptr;
            serial = 0;
            malformed = false;
            raisedh = nullptr;
            rho.clear();
        }
        bool same_raised(const vmn_garray* rh, const uint8_t* rho_be, size_t rho_bytes) const {
            return raisedh == rh && rho.size() == (rho_be ? rho_bytes : 0) && (rho.empty() || memcmp(rho.data(), rho_be, rho.size()) == 0);
        }
    } prep;
    int verify_prepare(const vmn_msg* rep, const vmn_garray* raisedh, const uint8_t* rho_be, size_t rho_bytes) {
        VMN_TRACE("ccpos:verify_prepare");
        REQUIRE(have_commitment && have_ab, "verify_prepare needs computeAB and setCommitment");
        const vmn_msg::Item *ikA = item_of(rep, 0, VMN_ITEM_RING), *ikB = item_of(rep, 1, VMN_ITEM_RING),
                            *ikE = item_of(rep, 2, VMN_ITEM_RARRAY);
        REQUIRE(rep && rep->items.size() == 3 && ikA && ikB && ikE, "reply is not (k_A, k_B, k_E)");
        REQUIRE(ikB->count == width && vmn_rarray_size(ikE->ra) == N && ikA->width == G.xb, "reply items have the wrong shape");
        REQUIRE(raised == (raisedh != nullptr) && raised == (rho_be != nullptr), "raised / plain form must match computeAB");
        prep.clear();
        if (!ring_items_in_range({ikA, ikB})) return mark_malformed(prep, rep);
        Num k_A = G.ring_from(ikA->bytes.data());
        std::vector<Num> k_B;
        for (auto& bts : split(*ikB)) k_B.push_back(G.ring_from(bts.data()));
        Num rho;
        HostJobs jobs;                                                            // (after everything its jobs touch)
        pk_powers(jobs, pkey, k_B, prep.pkpow);                                   // beside the multi-exponentiation below
        int kE_bits = 0;
        TRY(received_bits(ikE->ra, &kE_bits));
        if (!raised) {                                                            // :554-570
            jobs.start([&] { return gexp(g, k_A, prep.gkA); });
            std::vector<const vmn_garray*> xs{h};
            xs.insert(xs.end(), wp.begin(), wp.end());
            PendingProds kE_pending;
            TRY(kE_pending.begin(xs, ikE->ra, kE_bits));
            TRY(finish_ab());
            TRY(kE_pending.finish(G, prep.kE_prods));
            Round rd(*this);
            rd.products(prep.kE_prods);
            TRY(rd.run());
        } else {                                                                  // raised, single-equation form :571-580
            GA rh_own;
            const vmn_garray* rh = nullptr;
            REQUIRE(rho_bytes > 0, "empty raised exponent");
            rho = G.reduce(rho_be, rho_bytes);
            jobs.start([&] { return gexp(g, G.Zq.mul(k_A, rho), prep.g_term); });
            jobs.start([&] { return G.el_exp(cAp, rho_be, rho_bytes, prep.Ap_rho); });
            TRY(local_garray(raisedh, rh_own, &rh, "raised generators"));
            std::vector<GA> tmp(2 * width);
            std::vector<const vmn_garray*> xs;
            for (size_t c = 0; c < 2 * width; ++c) {
                TRY(vmn_garray_mul(wp[c], rh, tmp[c].out()));
                xs.push_back(tmp[c]);
            }
            PendingProds kE_pending;
            TRY(kE_pending.begin(xs, ikE->ra, kE_bits));
            TRY(finish_ab());
            TRY(kE_pending.finish(G, prep.prods));
            Round rd(*this);
            rd.products(prep.prods);
            TRY(rd.run());
        }
        TRY(jobs.join());
        prep.rep = rep;
        prep.serial = rep->serial;
        prep.epoch = epoch;
        prep.raisedh = raisedh;
        if (rho_be) prep.rho.assign(rho_be, rho_be + rho_bytes);
        return VMN_OK;
    }
    int verify(const vmn_msg* rep, const vmn_garray* raisedh, const uint8_t* rho_be, size_t rho_bytes, int* verdict) {
        VMN_TRACE("ccpos:verify");
        REQUIRE(verdict && have_commitment && have_ab && !v_be.empty(), "verify needs computeAB, setCommitment and setChallenge");
        *verdict = 0;
        if (!rep || prep.rep != rep || prep.serial != rep->serial || prep.epoch != epoch || !prep.same_raised(raisedh, rho_be, rho_bytes))
            TRY(verify_prepare(rep, raisedh, rho_be, rho_bytes));
        TRY(finish_ab());
        REQUIRE(raised == (raisedh != nullptr) && raised == (rho_be != nullptr), "raised / plain form must match computeAB");
        if (prep.malformed) {                                                     // a ring scalar >= q: not a reply
            prep.clear();
            return VMN_OK;                                                        // (*verdict = 0 above)
        }
        Bytes rhs, lhsA;
        std::vector<Bytes> rB, lhsB(2 * width);
        if (!raised) {                                                            // A^v A' = g^{k_A} prod h^{k_E}; B^v B' = pk^{-k_B} prod w'^{k_E}
            {
                HostJobs jobs;
                jobs.start([&] { return G.el_expmul(A, v_be, cAp, lhsA); });
                for (size_t c = 0; c < 2 * width; ++c) jobs.start([this, c, &lhsB] { return G.el_expmul(B[c], v_be, cBp[c], lhsB[c]); });
                TRY(jobs.join());
            }
            TRY(G.el_mul(prep.gkA, prep.kE_prods[0], rhs));
            const bool a_ok = lhsA == rhs;
            std::vector<Bytes> prods(prep.kE_prods.begin() + 1, prep.kE_prods.end());
            TRY(pk_finish(prep.pkpow, prods, rB));
            int ok = a_ok ? 1 : 0;                                                // :562 returns at the first failing check
            for (size_t c = 0; c < 2 * width; ++c) ok = ok && lhsB[c] == rB[c];
            prep.clear();
            *verdict = ok;
            return VMN_OK;
        }
        // raised:  AB^v (B' A'^rho) = pk^{-k_B} prod (w'_i h_i^rho)^{k_E,i} g^{k_A rho}
        {
            HostJobs jobs;
            for (size_t c = 0; c < 2 * width; ++c) {                              // AB_c^v (B'_c A'^rho): one job per component
                jobs.start([this, c, &lhsB] {
                    Bytes bt;
                    TRY(G.el_mul(cBp[c], prep.Ap_rho, bt));
                    return G.el_expmul(AB[c], v_be, bt, lhsB[c]);
                });
            }
            TRY(pk_finish(prep.pkpow, prep.prods, rB));
            TRY(jobs.join());
        }
        int ok = 1;
        for (size_t c = 0; c < 2 * width; ++c) {
            TRY(G.el_mul(rB[c], prep.g_term, rhs));
            ok = ok && lhsB[c] == rhs;
        }
        prep.clear();
        *verdict = ok;
        return VMN_OK;
    }
};


// ================================================================================================================
// Verifiable threshold decryption: DistrElGamalSession / DistrElGamalSessionBasic
// ================================================================================================================
namespace vmnp {

Num small(const HostGroup& G, uint64_t v) {
    Num r(G.ql, 0);
    r[0] = v;
    return r;
}
// c = (prod over primes p <= k of the largest power of p not exceeding k)^2 mod q   (:294-344)
Num prod_factor(const HostGroup& G, int k) {
    static const int primes[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97};
    Num res = G.Zq.reduce((const uint8_t*)"\x01", 1);
    for (int pr : primes) {
        if (pr > k) break;
        uint64_t a = 1, b = 1;
        while (b <= (uint64_t)k) {
            a = b;
            b *= (uint64_t)pr;
        }
        uint8_t be[8];
        for (int i = 0; i < 8; ++i) be[i] = (uint8_t)(a >> (8 * (7 - i)));
        res = G.Zq.mul(res, G.Zq.reduce(be, 8));
    }
    return G.Zq.mul(res, res);
}
// modified Lagrange coefficients of smallest absolute value (:358-452)
int lagrange(const HostGroup& G, const uint8_t* correct, int k, int threshold, std::vector<Num>& abs, std::vector<int>& negative,
             std::vector<int>* parties = nullptr) {
    const Num pf = prod_factor(G, k);
    abs.clear();
    negative.clear();
    auto num = [&](int v) {                         // small signed integer mod q
        uint8_t be[8];
        uint64_t a = (uint64_t)(v < 0 ? -v : v);
        for (int i = 0; i < 8; ++i) be[i] = (uint8_t)(a >> (8 * (7 - i)));
        Num r = G.Zq.reduce(be, 8);
        return v < 0 ? G.Zq.neg(r) : r;
    };
    for (int i = 1; i <= k && (int)abs.size() < threshold; ++i) {
        if (!correct[i]) continue;
        Num res = pf;
        int t = 0;
        for (int l = 1; l <= k && t < threshold; ++l) {
            if (!correct[l]) continue;
            if (l != i) res = G.Zq.mul(G.Zq.mul(res, num(l)), G.Zq.inv(num(l - i)));
            ++t;
        }
        Num alt = G.Zq.neg(res);                    // q - res = |res - q|
        const bool neg = !vmn::num64::is_zero(res) && vmn::num64::cmp(alt, res) < 0;
        abs.push_back(neg ? alt : res);
        negative.push_back(neg ? 1 : 0);
        if (parties) parties->push_back(i);
    }
    if ((int)abs.size() < threshold) return fail(VMN_ERR_ARG, "attempting to combine too few decryption factors");
    return VMN_OK;
}

}  // namespace vmnp

struct vmn_decproof {
    HostGroup G;
    int j = 0, k = 0, threshold = 0, ebitlen = 0, e_bits = 0;
    bool has_rs = false;
    vmn_random_source rs{};
    Num inverseFactor;
    const vmn_garray* u = nullptr;
    std::vector<Bytes> y;
    std::vector<const vmn_garray*> f;
    RA e;
    Bytes A;
    Num x, r;
    std::vector<Bytes> yp, Bp, B;
    std::vector<Num> k_x;
    std::vector<char> have_kx;
    // combined
    Bytes combinedyp, combinedBp, combinedy, combinedB;
    Num combinedk_x;
    const vmn_garray* combinedf = nullptr;

    int init(vmn_group* grp, int j_, int k_, int threshold_, int ebitlen_, const vmn_random_source* r_) {
        TRY(G.init(grp));
        j = j_;
        k = k_;
        threshold = threshold_;
        ebitlen = ebitlen_;
        e_bits = std::min(ebitlen_, G.qbits);
        if (r_) {
            if (!r_->ring_elements) return fail(VMN_ERR_ARG, "random source lacks a callback");
            rs = *r_;
            has_rs = true;
        }
        inverseFactor = G.Zq.inv(prod_factor(G, k));
        y.assign(k + 1, Bytes());
        f.assign(k + 1, nullptr);
        yp.assign(k + 1, Bytes());
        Bp.assign(k + 1, Bytes());
        B.assign(k + 1, Bytes());
        k_x.assign(k + 1, Num());
        have_kx.assign(k + 1, 0);
        return VMN_OK;
    }
    int party(int l) const { return l >= 1 && l <= k ? VMN_OK : fail(VMN_ERR_ARG, "party index %d outside 1..%d", l, k); }
    int set_instance(const vmn_garray* u_, const uint8_t* y_be, const vmn_garray* const* f_) {
        REQUIRE(u_ && y_be && f_, "null argument");
        u = u_;
        for (int l = 1; l <= k; ++l) {
            y[l].assign(y_be + (size_t)l * G.eb, y_be + (size_t)(l + 1) * G.eb);
            f[l] = f_[l];
            if (f[l] && vmn_garray_size(f[l]) != vmn_garray_size(u)) return fail(VMN_ERR_ARG, "decryption factors of party %d differ in size from u", l);
        }
        return VMN_OK;
    }
    int batch_input() {
        REQUIRE(u && e.p, "batchInput needs the instance and the batching vector");
        A.resize(G.eb);
        return vmn_garray_expprod(u, e, e_bits, A.data());                       // :524-526
    }
    int commit(const uint8_t* x_be, uint8_t* yp_out, uint8_t* Bp_out) {
        REQUIRE(x_be && yp_out && Bp_out && has_rs && !A.empty(), "commit needs a random source and batchInput()");
        x = G.ring_from(x_be);
        const uint8_t* rows = nullptr;
        if (rs.ring_elements(rs.user, 1, &rows) != 0 || !rows) return fail(VMN_ERR_ARG, "random source failed");
        r = G.ring_from(rows);
        if (vmn::num64::cmp(r, G.Zq.n) >= 0) return fail(VMN_ERR_FORMAT, "random source returned a value >= q");
        TRY(G.el_exp(G.g, r, yp[j]));                                            // y' = g^r
        TRY(G.el_exp(A, r, Bp[j]));                                              // B' = A^r
        memcpy(yp_out, yp[j].data(), G.eb);
        memcpy(Bp_out, Bp[j].data(), G.eb);
        return VMN_OK;
    }
    int reply(const uint8_t* v_be, size_t vbytes, uint8_t* kx_out) {
        REQUIRE(v_be && vbytes && kx_out && !x.empty() && !r.empty(), "reply needs commit()");
        Num v = G.reduce(v_be, vbytes);
        k_x[j] = G.Zq.add(G.Zq.mul(G.Zq.mul(G.Zq.neg(x), inverseFactor), v), r);  // -x c^-1 v + r   :595-598
        have_kx[j] = 1;
        Bytes out = G.ring_bytes(k_x[j]);
        memcpy(kx_out, out.data(), G.xb);
        return VMN_OK;
    }
    int verify(int l, const uint8_t* v_be, size_t vbytes, int* verdict) {
        TRY(party(l));
        REQUIRE(verdict && v_be && vbytes && !A.empty() && !B[l].empty() && !yp[l].empty() && have_kx[l], "verify needs batch(l), the commitment and the reply of l");
        if (have_kx[l] == 2) {                                                   // malformed reply: verdicts[l] = false (:719-721)
            *verdict = 0;
            return VMN_OK;
        }
        Num v = G.reduce(v_be, vbytes);
        Bytes yinv, t, lhs, rhs;
        TRY(G.el_inv(y[l], yinv));
        TRY(G.el_exp(yinv, G.Zq.mul(inverseFactor, v), t));
        TRY(G.el_mul(t, yp[l], lhs));
        TRY(G.el_exp(G.g, k_x[l], rhs));
        const int ok1 = lhs == rhs;
        TRY(G.el_exp(B[l], v, t));
        TRY(G.el_mul(t, Bp[l], lhs));
        TRY(G.el_exp(A, k_x[l], rhs));
        *verdict = ok1 && lhs == rhs;
        return VMN_OK;
    }
    int combine(const uint8_t* correct, const uint8_t* combinedy_be, const vmn_garray* combinedf_) {
        REQUIRE(correct && combinedy_be && combinedf_, "null argument");
        std::vector<Num> abs;
        std::vector<int> neg, parties;
        TRY(lagrange(G, correct, k, threshold, abs, neg, &parties));
        combinedyp = G.one();
        combinedBp = G.one();
        combinedk_x = Num(G.ql, 0);
        for (size_t t = 0; t < parties.size(); ++t) {
            const int l = parties[t];
            REQUIRE(!yp[l].empty() && have_kx[l], "combine needs the commitment and the reply of every combined party");
            Num ex = neg[t] ? G.Zq.neg(abs[t]) : abs[t];
            Bytes a, b2;
            TRY(G.el_exp(yp[l], ex, a));
            TRY(G.el_mul(combinedyp, a, b2));
            combinedyp = b2;
            TRY(G.el_exp(Bp[l], ex, a));
            TRY(G.el_mul(combinedBp, a, b2));
            combinedBp = b2;
            combinedk_x = G.Zq.add(combinedk_x, G.Zq.mul(k_x[l], ex));
        }
        combinedy.assign(combinedy_be, combinedy_be + G.eb);
        combinedf = combinedf_;
        return VMN_OK;
    }
    int batch_combined() {
        REQUIRE(combinedf && e.p, "batchCombined needs combine() and the batching vector");
        combinedB.resize(G.eb);
        return vmn_garray_expprod(combinedf, e, e_bits, combinedB.data());
    }
    int verify_combined(const uint8_t* v_be, size_t vbytes, int* verdict) {
        REQUIRE(verdict && v_be && vbytes && !combinedB.empty() && !A.empty(), "verifyCombined needs batchCombined()");
        Num v = G.reduce(v_be, vbytes);
        Bytes yinv, t, lhs, rhs;
        TRY(G.el_inv(combinedy, yinv));
        TRY(G.el_exp(yinv, v, t));
        TRY(G.el_mul(t, combinedyp, lhs));
        TRY(G.el_exp(G.g, combinedk_x, rhs));
        const int ok1 = lhs == rhs;
        TRY(G.el_exp(combinedB, v, t));
        TRY(G.el_mul(t, combinedBp, lhs));
        TRY(G.el_exp(A, combinedk_x, rhs));
        *verdict = ok1 && lhs == rhs;
        return VMN_OK;
    }
};


// ================================================================================================================
// IndependentGeneratorsBasicI (interactive derivation of independent generators)
// ================================================================================================================
struct vmn_igen {
    HostGroup G;
    int j = 0, threshold = 0, ebitlen = 0, e_bits = 0;
    bool has_rs = false;
    vmn_random_source rs{};
    Bytes g;
    std::vector<const vmn_garray*> h;
    const vmn_rarray* s = nullptr;
    const vmn_garray* combinedh = nullptr;
    size_t N = 0;
    RA e;
    Num a, r, v;
    Bytes v_be;
    std::vector<Bytes> Ap;
    std::vector<Num> k_a;
    std::vector<char> have;

    int init(vmn_group* grp, int j_, int threshold_, int ebitlen_, const vmn_random_source* r_) {
        TRY(G.init(grp));
        j = j_;
        threshold = threshold_;
        ebitlen = ebitlen_;
        e_bits = std::min(ebitlen_, G.qbits);
        if (r_) {
            if (!r_->ring_elements) return fail(VMN_ERR_ARG, "random source lacks a callback");
            rs = *r_;
            has_rs = true;
        }
        h.assign(threshold + 1, nullptr);
        Ap.assign(threshold + 1, Bytes());
        k_a.assign(threshold + 1, Num());
        have.assign(threshold + 1, 0);
        return VMN_OK;
    }
    int party(int l) const { return l >= 1 && l <= threshold ? VMN_OK : fail(VMN_ERR_ARG, "party index %d outside 1..%d", l, threshold); }
    int commit(uint8_t* out) {
        REQUIRE(out && s && e.p && has_rs, "commit needs the exponents s, the batching vector and a random source");
        Bytes ab(G.xb);
        TRY(vmn_rarray_inner_product(s, e, ab.data()));                            // a = <s, e>   :202
        a = G.ring_from(ab.data());
        const uint8_t* rows = nullptr;
        if (rs.ring_elements(rs.user, 1, &rows) != 0 || !rows) return fail(VMN_ERR_ARG, "random source failed");
        r = G.ring_from(rows);
        if (vmn::num64::cmp(r, G.Zq.n) >= 0) return fail(VMN_ERR_FORMAT, "random source returned a value >= q");
        TRY(G.el_exp(g, r, Ap[j]));                                                // A' = g^r     :205
        memcpy(out, Ap[j].data(), G.eb);
        return VMN_OK;
    }
    int check(const vmn_garray* harr, const Bytes& Apl, const Num& kal, int* verdict) {
        Bytes A(G.eb), t, lhs, rhs;
        TRY(vmn_garray_expprod(harr, e, e_bits, A.data()));
        TRY(G.el_exp(A, v, t));
        TRY(G.el_mul(t, Apl, lhs));
        TRY(G.el_exp(g, kal, rhs));
        *verdict = lhs == rhs;
        return VMN_OK;
    }
};

// ================================================================================================================
// C entry points
// ================================================================================================================
extern "C" {

int vmn_msg_create(vmn_msg** out) {
    if (!out) return fail(VMN_ERR_ARG, "vmn_msg_create: null argument");
    *out = new vmn_msg();
    return VMN_OK;
}
int vmn_msg_create_for(vmn_group* grp, vmn_msg** out) {
    if (!out || !grp) return fail(VMN_ERR_ARG, "vmn_msg_create_for: null argument");
    *out = new vmn_msg(vmn_group_kind(grp) == 1);
    return VMN_OK;
}
void vmn_msg_free(vmn_msg* m) { delete m; }
size_t vmn_msg_items(const vmn_msg* m) { return m ? m->items.size() : 0; }
int vmn_msg_item_kind(const vmn_msg* m, size_t i) { return m && i < m->items.size() ? m->items[i].kind : 0; }
const vmn_garray* vmn_msg_item_garray(const vmn_msg* m, size_t i) {
    const vmn_msg::Item* it = item_of(m, i, VMN_ITEM_GARRAY);
    return it ? it->ga : nullptr;
}
const vmn_rarray* vmn_msg_item_rarray(const vmn_msg* m, size_t i) {
    const vmn_msg::Item* it = item_of(m, i, VMN_ITEM_RARRAY);
    return it ? it->ra : nullptr;
}
int vmn_msg_item_bytes(const vmn_msg* m, size_t i, const uint8_t** data, size_t* count, size_t* width) {
    if (!m || i >= m->items.size() || (m->items[i].kind != VMN_ITEM_ELEMENTS && m->items[i].kind != VMN_ITEM_RING))
        return fail(VMN_ERR_ARG, "vmn_msg_item_bytes: item %zu is not a scalar item", i);
    if (data) *data = m->items[i].bytes.data();
    if (count) *count = m->items[i].count;
    if (width) *width = m->items[i].width;
    return VMN_OK;
}
int vmn_msg_push_garray(vmn_msg* m, vmn_garray* a) {
    if (!m || !a) return fail(VMN_ERR_ARG, "vmn_msg_push_garray: null argument");
    vmn_msg::Item it;
    it.kind = VMN_ITEM_GARRAY;
    it.ga = a;
    m->items.push_back(std::move(it));
    return VMN_OK;
}
int vmn_msg_push_rarray(vmn_msg* m, vmn_rarray* a) {
    if (!m || !a) return fail(VMN_ERR_ARG, "vmn_msg_push_rarray: null argument");
    vmn_msg::Item it;
    it.kind = VMN_ITEM_RARRAY;
    it.ra = a;
    m->items.push_back(std::move(it));
    return VMN_OK;
}
static int push_scalar_item(vmn_msg* m, int kind, const uint8_t* be, size_t count, size_t width) {
    if (!m || (!be && count) || !width) return fail(VMN_ERR_ARG, "vmn_msg_push: null argument");
    vmn_msg::Item it;
    it.kind = kind;
    it.count = count;
    it.width = width;
    it.bytes.assign(be, be + count * width);
    m->items.push_back(std::move(it));
    return VMN_OK;
}
int vmn_msg_push_elements(vmn_msg* m, const uint8_t* be, size_t count, size_t width) {
    return push_scalar_item(m, VMN_ITEM_ELEMENTS, be, count, width);
}
int vmn_msg_push_ring(vmn_msg* m, const uint8_t* be, size_t count, size_t width) {
    return push_scalar_item(m, VMN_ITEM_RING, be, count, width);
}

// ---- wire form -------------------------------------------------------------------------------------------------
static void put_header(uint8_t*& o, uint8_t tag, size_t n) {
    *o++ = tag;
    *o++ = (uint8_t)(n >> 24);
    *o++ = (uint8_t)(n >> 16);
    *o++ = (uint8_t)(n >> 8);
    *o++ = (uint8_t)n;
}
static size_t scalar_item_size(const vmn_msg::Item& it, bool ec) {
    // one element: a leaf, or -- a curve point -- node(leaf(x), leaf(y))
    const size_t leaf = (ec && it.kind == VMN_ITEM_ELEMENTS) ? 15 + it.width : 5 + it.width;
    if (it.count == 1) return leaf;
    if (it.kind == VMN_ITEM_RING) return 5 + it.count * leaf;
    if (it.count == 2) return 5 + 2 * leaf;
    return 5 + 2 * (5 + (it.count / 2) * leaf);
}
size_t vmn_msg_bytetree_size(const vmn_msg* m) {
    if (!m) return 0;
    size_t total = 5;
    for (auto& it : m->items) {
        if (it.kind == VMN_ITEM_GARRAY) total += vmn_garray_bytetree_size(it.ga);
        else if (it.kind == VMN_ITEM_RARRAY) total += vmn_rarray_bytetree_size(it.ra);
        else total += scalar_item_size(it, m->ec);
    }
    return total;
}
int vmn_msg_to_bytetree(const vmn_msg* m, uint8_t* out) {
    if (!m || !out) return fail(VMN_ERR_ARG, "vmn_msg_to_bytetree: null argument");
    uint8_t* o = out;
    put_header(o, 0, m->items.size());
    for (auto& it : m->items) {
        if (it.kind == VMN_ITEM_GARRAY) {
            TRY(vmn_garray_to_bytetree(it.ga, o));
            o += vmn_garray_bytetree_size(it.ga);
            continue;
        }
        if (it.kind == VMN_ITEM_RARRAY) {
            TRY(vmn_rarray_to_bytetree(it.ra, o));
            o += vmn_rarray_bytetree_size(it.ra);
            continue;
        }
        auto leaf = [&](size_t k) {
            const uint8_t* src = it.bytes.data() + k * it.width;
            if (m->ec && it.kind == VMN_ITEM_ELEMENTS) {            // a curve point: node(leaf(x), leaf(y))
                const size_t cb = it.width / 2;
                put_header(o, 0, 2);
                for (int half = 0; half < 2; ++half) {
                    put_header(o, 1, cb);
                    memcpy(o, src + half * cb, cb);
                    o += cb;
                }
                return;
            }
            put_header(o, 1, it.width);
            memcpy(o, src, it.width);
            o += it.width;
        };
        if (it.count == 1) {
            leaf(0);
        } else if (it.kind == VMN_ITEM_RING || it.count == 2) {
            put_header(o, 0, it.count);
            for (size_t k = 0; k < it.count; ++k) leaf(k);
        } else {
            put_header(o, 0, 2);
            for (size_t half = 0; half < 2; ++half) {
                put_header(o, 0, it.count / 2);
                for (size_t k = 0; k < it.count / 2; ++k) leaf(half * (it.count / 2) + k);
            }
        }
    }
    return VMN_OK;
}

namespace {
struct Reader {
    const uint8_t* p;
    const uint8_t* end;
    bool header(uint8_t tag, size_t* n) {
        if (end - p < 5 || p[0] != tag) return false;
        *n = ((size_t)p[1] << 24) | ((size_t)p[2] << 16) | ((size_t)p[3] << 8) | p[4];
        p += 5;
        return true;
    }
    bool leaf(size_t width, Bytes& sink) {
        size_t n;
        if (!header(1, &n) || n != width || (size_t)(end - p) < width) return false;
        sink.insert(sink.end(), p, p + width);
        p += width;
        return true;
    }
    // one group element: a leaf, or -- a curve point -- node(leaf(x), leaf(y)) appended as x || y
    bool element(size_t width, bool point, Bytes& sink) {
        if (!point) return leaf(width, sink);
        size_t n;
        return header(0, &n) && n == 2 && leaf(width / 2, sink) && leaf(width / 2, sink);
    }
};
}  // namespace

int vmn_msg_from_bytetree(vmn_group* grp, const uint8_t* bt, size_t len, const int* layout, const size_t* counts, size_t items,
                          vmn_msg** out, int* format_ok) {
    if (!grp || !bt || !layout || !counts || !out || !format_ok) return fail(VMN_ERR_ARG, "vmn_msg_from_bytetree: null argument");
    *format_ok = 0;
    *out = nullptr;
    const size_t eb = vmn_group_elem_bytes(grp), xb = vmn_group_exp_bytes(grp);
    Reader rd{bt, bt + len};
    size_t n;
    if (!rd.header(0, &n) || n != items) return VMN_OK;
    const bool ec = vmn_group_kind(grp) == 1;
    std::unique_ptr<vmn_msg> m(new vmn_msg(ec));
    for (size_t i = 0; i < items; ++i) {
        if (layout[i] == VMN_ITEM_GARRAY || layout[i] == VMN_ITEM_RARRAY) {
            const size_t width = layout[i] == VMN_ITEM_GARRAY ? eb : xb;
            const size_t need = 5 + counts[i] * ((ec && layout[i] == VMN_ITEM_GARRAY) ? 15 + width : 5 + width);
            if ((size_t)(rd.end - rd.p) < need) return VMN_OK;
            int ok = 0, in_range = 1;
            if (layout[i] == VMN_ITEM_GARRAY) {
                vmn_garray* a = nullptr;
                TRY(vmn_garray_from_bytetree(grp, rd.p, need, counts[i], &a, &ok, &in_range));
                if (a) vmn_msg_push_garray(m.get(), a);
                if (a && ok && in_range) {                 // pGroup.toElementArray also checks subgroup membership
                    int member = 1;
                    TRY(vmn_garray_is_member(a, &member));
                    if (!member) in_range = 0;
                }
            } else {
                vmn_rarray* a = nullptr;
                TRY(vmn_rarray_from_bytetree(grp, rd.p, need, counts[i], &a, &ok, &in_range));
                if (a) vmn_msg_push_rarray(m.get(), a);
            }
            if (!ok || !in_range) return VMN_OK;
            rd.p += need;
            continue;
        }
        vmn_msg::Item it;
        it.kind = layout[i];
        it.count = counts[i];
        it.width = layout[i] == VMN_ITEM_ELEMENTS ? eb : xb;
        bool good = true;
        const bool point = ec && it.kind == VMN_ITEM_ELEMENTS;
        if (it.count == 1) {
            good = rd.element(it.width, point, it.bytes);
        } else if (it.kind == VMN_ITEM_RING || it.count == 2) {
            good = rd.header(0, &n) && n == it.count;
            for (size_t k = 0; good && k < it.count; ++k) good = rd.element(it.width, point, it.bytes);
        } else {
            good = rd.header(0, &n) && n == 2;
            for (size_t half = 0; good && half < 2; ++half) {
                good = rd.header(0, &n) && n == it.count / 2;
                for (size_t k = 0; good && k < it.count / 2; ++k) good = rd.element(it.width, point, it.bytes);
            }
        }
        if (!good) return VMN_OK;
        if (it.kind == VMN_ITEM_RING) {                // pRing.toElement rejects a value >= q (PoSBasicTW.java:985-989)
            Bytes qb(xb);
            TRY(vmn_group_get_order(grp, qb.data()));
            for (size_t k = 0; k < it.count; ++k)
                if (memcmp(it.bytes.data() + k * xb, qb.data(), xb) >= 0) return VMN_OK;
        }
        m->items.push_back(std::move(it));
    }
    if (rd.p != rd.end) return VMN_OK;
    *format_ok = 1;
    *out = m.release();
    return VMN_OK;
}

// ---- shuffler lines --------------------------------------------------------------------------------------------
int vmn_shuffle_reencrypt(vmn_group* grp, const uint8_t* pkey_be, size_t width, const vmn_garray* const* w,
                          const vmn_rarray* const* s, const uint32_t* pi, vmn_garray** wp_out) {
    if (!grp || !pkey_be || !width || !w || !s || !pi || !wp_out) return fail(VMN_ERR_ARG, "vmn_shuffle_reencrypt: null argument");
    const size_t eb = vmn_group_elem_bytes(grp);
    const size_t n = vmn_garray_size(w[0]);
    if (!is_permutation(pi, n)) return fail(VMN_ERR_ARG, "vmn_shuffle_reencrypt: pi is not a permutation of [0, N)");
    std::vector<uint32_t> inv = inverse_permutation(pi, n);
    std::vector<GA> res(2 * width);
    for (size_t c = 0; c < 2 * width; ++c) {
        GA factors, reenc;                                                        // freed right after use (:274)
        TRY(vmn_group_exp_fixed(grp, pkey_be + c * eb, s[c % width], factors.out()));     // widePublicKey.exp(reencExponents) :407
        TRY(vmn_garray_mul(w[c], factors, reenc.out()));                          // input.mul(reencFactors) :273
        TRY(vmn_garray_permute(reenc, inv.data(), res[c].out()));                 // permute(permutation.inv()) :278
    }
    for (size_t c = 0; c < 2 * width; ++c) wp_out[c] = res[c].release();
    return VMN_OK;
}

// The same in the two steps of the reference's precomputed shuffle: the factors in `vmn -precomp`
// (ShufflerElGamalSession.java:645-661: reencFactors = widePublicKey.exp(reencExponents)), the rest when the ciphertexts
// arrive (:789-792: input.mul(reencFactors), permute(permutation.inv())).
int vmn_shuffle_reencryption_factors(vmn_group* grp, const uint8_t* pkey_be, size_t width, const vmn_rarray* const* s,
                                     vmn_garray** factors_out) {
    if (!grp || !pkey_be || !width || !s || !factors_out) return fail(VMN_ERR_ARG, "vmn_shuffle_reencryption_factors: null argument");
    const size_t eb = vmn_group_elem_bytes(grp);
    std::vector<GA> res(2 * width);
    for (size_t c = 0; c < 2 * width; ++c) TRY(vmn_group_exp_fixed(grp, pkey_be + c * eb, s[c % width], res[c].out()));
    for (size_t c = 0; c < 2 * width; ++c) factors_out[c] = res[c].release();
    return VMN_OK;
}
int vmn_shuffle_apply_factors(vmn_group* grp, size_t width, const vmn_garray* const* w, const vmn_garray* const* factors,
                              const uint32_t* pi, vmn_garray** wp_out) {
    if (!grp || !width || !w || !factors || !pi || !wp_out) return fail(VMN_ERR_ARG, "vmn_shuffle_apply_factors: null argument");
    for (size_t c = 0; c < 2 * width; ++c)
        if (!w[c] || !factors[c] || vmn_garray_size(w[c]) != vmn_garray_size(w[0]) || vmn_garray_size(factors[c]) != vmn_garray_size(w[0]))
            return fail(VMN_ERR_ARG, "vmn_shuffle_apply_factors: arrays differ in size");
    const size_t n = vmn_garray_size(w[0]);
    if (!is_permutation(pi, n)) return fail(VMN_ERR_ARG, "vmn_shuffle_apply_factors: pi is not a permutation of [0, N)");
    std::vector<uint32_t> inv = inverse_permutation(pi, n);
    std::vector<GA> res(2 * width);
    for (size_t c = 0; c < 2 * width; ++c) {
        GA reenc;
        TRY(vmn_garray_mul(w[c], factors[c], reenc.out()));                       // input.mul(reencFactors) :789
        TRY(vmn_garray_permute(reenc, inv.data(), res[c].out()));                 // permute(permutation.inv()) :792
    }
    for (size_t c = 0; c < 2 * width; ++c) wp_out[c] = res[c].release();
    return VMN_OK;
}

int vmn_permutation_commitment(vmn_group* grp, const uint8_t* g_be, const vmn_garray* h, const vmn_rarray* r, const uint32_t* pi,
                               vmn_garray** u_out) {
    if (!grp || !g_be || !h || !r || !pi || !u_out) return fail(VMN_ERR_ARG, "vmn_permutation_commitment: null argument");
    if (!is_permutation(pi, vmn_garray_size(h))) return fail(VMN_ERR_ARG, "vmn_permutation_commitment: pi is not a permutation of [0, N)");
    GA tmp1, tmp2;
    TRY(vmn_group_exp_fixed(grp, g_be, r, tmp1.out()));                           // g.exp(exponents) :200
    TRY(vmn_garray_mul(h, tmp1, tmp2.out()));                                     // generators.mul(tmp) :201
    tmp1.reset();
    return vmn_garray_permute(tmp2, pi, u_out);                                   // :215
}

void vmn_shard_bounds(size_t n, int world, int rank, size_t* lo, size_t* hi) {
    const size_t base = n / (size_t)world, rem = n % (size_t)world, k = (size_t)rank;
    const size_t a = k * base + (k < rem ? k : rem);
    if (lo) *lo = a;
    if (hi) *hi = a + base + (k < rem ? 1 : 0);
}

// this rank's positions [lo, hi) of u = permute(h g^r, pi): u_i = h_pi(i) g^(r_pi(i)), gathered out of the whole h and r
int vmn_permutation_commitment_shard(vmn_group* grp, const uint8_t* g_be, const vmn_garray* h_full, const vmn_rarray* r_full,
                                     const uint32_t* pi, size_t lo, size_t hi, vmn_garray** u_out) {
    if (!grp || !g_be || !h_full || !r_full || !pi || !u_out || lo > hi || hi > vmn_garray_size(h_full) ||
        vmn_rarray_size(r_full) != vmn_garray_size(h_full))
        return fail(VMN_ERR_ARG, "vmn_permutation_commitment_shard: bad argument");
    RA rsel;
    TRY(vmn_rarray_gather(r_full, pi + lo, hi - lo, rsel.out()));
    return vmnp::permutation_commitment_rows(grp, g_be, h_full, rsel, pi + lo, hi - lo, u_out);
}

// this rank's positions [lo, hi) of w' = permute(w pk^s, pi^-1), gathered out of the whole w and s
int vmn_shuffle_reencrypt_shard(vmn_group* grp, const uint8_t* pkey_be, size_t width, const vmn_garray* const* w_full,
                                const vmn_rarray* const* s_full, const uint32_t* pi, size_t lo, size_t hi, vmn_garray** wp_out) {
    if (!grp || !pkey_be || !width || !w_full || !s_full || !pi || !wp_out || lo > hi) return fail(VMN_ERR_ARG, "vmn_shuffle_reencrypt_shard: null argument");
    const size_t n = vmn_garray_size(w_full[0]);
    if (hi > n || !is_permutation(pi, n)) return fail(VMN_ERR_ARG, "vmn_shuffle_reencrypt_shard: pi is not a permutation of [0, N) or the range is outside it");
    std::vector<uint32_t> inv = inverse_permutation(pi, n);
    std::vector<RA> ssel(width);
    std::vector<const vmn_rarray*> sp(width);
    for (size_t c = 0; c < width; ++c) {
        TRY(vmn_rarray_gather(s_full[c], inv.data() + lo, hi - lo, ssel[c].out()));
        sp[c] = ssel[c];
    }
    return vmnp::reencrypt_rows(grp, pkey_be, width, w_full, sp.data(), inv.data() + lo, hi - lo, wp_out);
}

// The same two lines for a party whose exponents are PRG draws (vmn_random_source.array_seed): the rank generates the rows
// it needs -- s_{pi^-1(i)} resp. r_{pi(i)} for its positions, and its own positions of s resp. r (what the proof object is
// handed afterwards) -- instead of the whole arrays.  The source is asked for the same seeds in the same order as
// vmn_rarray_random would ask (one per column), so every rank -- and a single-GPU run on the same source -- computes
// the same shuffle.
int vmn_shuffle_reencrypt_shard_seeded(vmn_group* grp, const uint8_t* pkey_be, size_t width, const vmn_garray* const* w_full,
                                       const vmn_random_source* rs, int rbitlen, const uint32_t* pi, size_t lo, size_t hi,
                                       vmn_garray** wp_out, vmn_rarray** s_out) {
    if (!grp || !pkey_be || !width || !w_full || !rs || !pi || !wp_out || !s_out || lo > hi || rbitlen < 0)
        return fail(VMN_ERR_ARG, "vmn_shuffle_reencrypt_shard_seeded: bad argument");
    const size_t n = vmn_garray_size(w_full[0]);
    if (hi > n || !is_permutation(pi, n)) return fail(VMN_ERR_ARG, "vmn_shuffle_reencrypt_shard_seeded: pi is not a permutation of [0, N) or the range is outside it");
    HostGroup G;
    TRY(G.init(grp));
    std::vector<uint32_t> inv = inverse_permutation(pi, n);
    std::vector<RA> ssel(width), smine(width);
    std::vector<const vmn_rarray*> sp(width);
    for (size_t c = 0; c < width; ++c) {
        vmnp::Draw d;
        TRY(vmnp::make_draw(grp, *rs, n, true, G.qbits + rbitlen, d));
        TRY(d.rows(grp, inv.data() + lo, hi - lo, ssel[c]));
        TRY(d.range(grp, lo, hi, smine[c]));
        sp[c] = ssel[c];
    }
    TRY(vmnp::reencrypt_rows(grp, pkey_be, width, w_full, sp.data(), inv.data() + lo, hi - lo, wp_out));
    for (size_t c = 0; c < width; ++c) s_out[c] = smine[c].release();
    return VMN_OK;
}
int vmn_permutation_commitment_shard_seeded(vmn_group* grp, const uint8_t* g_be, const vmn_garray* h_full, const vmn_random_source* rs,
                                            int rbitlen, const uint32_t* pi, size_t lo, size_t hi, vmn_garray** u_out, vmn_rarray** r_out) {
    if (!grp || !g_be || !h_full || !rs || !pi || !u_out || !r_out || lo > hi || hi > vmn_garray_size(h_full) || rbitlen < 0)
        return fail(VMN_ERR_ARG, "vmn_permutation_commitment_shard_seeded: bad argument");
    const size_t n = vmn_garray_size(h_full);
    if (!is_permutation(pi, n)) return fail(VMN_ERR_ARG, "vmn_permutation_commitment_shard_seeded: pi is not a permutation of [0, N)");
    HostGroup G;
    TRY(G.init(grp));
    vmnp::Draw d;
    RA rsel, rmine;
    TRY(vmnp::make_draw(grp, *rs, n, true, G.qbits + rbitlen, d));
    TRY(d.rows(grp, pi + lo, hi - lo, rsel));
    TRY(d.range(grp, lo, hi, rmine));
    TRY(vmnp::permutation_commitment_rows(grp, g_be, h_full, rsel, pi + lo, hi - lo, u_out));
    *r_out = rmine.release();
    return VMN_OK;
}

int vmn_rarray_random(vmn_group* grp, const vmn_random_source* rs, size_t n, int rbitlen, vmn_rarray** out) {
    if (!grp || !rs || !out || rbitlen < 0 || (!rs->ring_elements && !rs->array_seed)) return fail(VMN_ERR_ARG, "vmn_rarray_random: bad argument");
    HostGroup G;
    TRY(G.init(grp));
    RA a;
    TRY(random_ring_array(grp, *rs, n, G.qbits, rbitlen, a));
    *out = a.release();
    return VMN_OK;
}

int vmn_permutation_shrink(const uint32_t* pi, size_t n_max, size_t n, uint8_t* keep_out, uint32_t* pi_out) {
    if (!pi || !keep_out || !pi_out || n > n_max) return fail(VMN_ERR_ARG, "vmn_permutation_shrink: bad argument");
    if (!is_permutation(pi, n_max)) return fail(VMN_ERR_ARG, "vmn_permutation_shrink: pi is not a permutation of [0, n_max)");
    size_t k = 0;
    for (size_t i = 0; i < n_max; ++i) {
        keep_out[i] = pi[i] < n ? 1 : 0;                                          // :398-405
        if (keep_out[i]) pi_out[k++] = pi[i];                                     // permutation.shrink(noCiphertexts)
    }
    return VMN_OK;
}
int vmn_keep_list_sanitize(uint8_t* keep, size_t keep_len, size_t n_max, size_t n, int* replaced) {
    if (!keep || n > n_max) return fail(VMN_ERR_ARG, "vmn_keep_list_sanitize: bad argument");
    size_t count = 0;
    bool trivial = keep_len != n_max;                                             // readBooleans(commitment.size()) failed
    for (size_t i = 0; !trivial && i < n_max; ++i) count += keep[i] ? 1 : 0;
    if (!trivial && count != n) trivial = true;                                   // :437-439
    if (trivial) {
        for (size_t i = 0; i < n_max; ++i) keep[i] = i < n ? 1 : 0;               // :442-445
    }
    if (replaced) *replaced = trivial ? 1 : 0;
    return VMN_OK;
}

// ---- proof objects ---------------------------------------------------------------------------------------------
#define CREATE(T, name)                                                                                              \
    int name(vmn_group* grp, int vbitlen, int ebitlen, int rbitlen, const vmn_random_source* rs, T** out) {          \
        if (!grp || !out || vbitlen <= 0 || ebitlen <= 0 || rbitlen < 0) return fail(VMN_ERR_ARG, #name ": bad argument"); \
        std::unique_ptr<T> p(new T());                                                                               \
        TRY(p->init(grp, vbitlen, ebitlen, rbitlen, rs));                                                            \
        *out = p.release();                                                                                          \
        return VMN_OK;                                                                                               \
    }
CREATE(vmn_pos, vmn_pos_create)
CREATE(vmn_posc, vmn_posc_create)
CREATE(vmn_ccpos, vmn_ccpos_create)
#undef CREATE
#define NONNULL(p) \
    if (!(p)) return fail(VMN_ERR_ARG, "%s: null proof object", __func__)

void vmn_pos_free(vmn_pos* p) {
    if (p) p->release_base_table();
    delete p;
}
vmn_group* vmn_pos_group(const vmn_pos* p) { return p ? p->G.grp : nullptr; }
vmn_group* vmn_posc_group(const vmn_posc* p) { return p ? p->G.grp : nullptr; }
vmn_group* vmn_ccpos_group(const vmn_ccpos* p) { return p ? p->G.grp : nullptr; }
vmn_group* vmn_decproof_group(const vmn_decproof* p) { return p ? p->G.grp : nullptr; }
vmn_group* vmn_igen_group(const vmn_igen* p) { return p ? p->G.grp : nullptr; }
size_t vmn_pos_size(const vmn_pos* p) { return p ? p->Ntot : 0; }
size_t vmn_posc_size(const vmn_posc* p) { return p ? p->Ntot : 0; }
size_t vmn_ccpos_size(const vmn_ccpos* p) { return p ? p->Ntot : 0; }
size_t vmn_decproof_size(const vmn_decproof* p) { return p ? vmn_garray_size(p->u) : 0; }
size_t vmn_igen_size(const vmn_igen* p) { return p ? p->N : 0; }
int vmn_decproof_parties(const vmn_decproof* p) { return p ? p->k : 0; }
int vmn_igen_parties(const vmn_igen* p) { return p ? p->threshold : 0; }
int vmn_pos_precompute(vmn_pos* p, const uint8_t* g_be, const vmn_garray* h, const uint32_t* pi) {
    NONNULL(p);
    return p->precompute(g_be, h, pi);
}
const vmn_garray* vmn_pos_permutation_commitment(const vmn_pos* p) { return p ? p->u : nullptr; }
int vmn_pos_set_permutation_commitment(vmn_pos* p, const vmn_garray* u) {
    NONNULL(p);
    return p->set_permutation_commitment(u);
}
int vmn_pos_set_comm(vmn_pos* p, const vmn_comm* comm) {
    NONNULL(p);
    return p->set_comm(comm);
}
int vmn_posc_set_comm(vmn_posc* p, const vmn_comm* comm) {
    NONNULL(p);
    return p->set_comm(comm);
}
int vmn_ccpos_set_comm(vmn_ccpos* p, const vmn_comm* comm) {
    NONNULL(p);
    return p->set_comm(comm);
}
int vmn_pos_set_instance(vmn_pos* p, const uint8_t* pkey_be, size_t width, const vmn_garray* const* w, const vmn_garray* const* wp,
                         const vmn_rarray* const* s) {
    NONNULL(p);
    return p->set_instance(pkey_be, width, w, wp, s);
}
int vmn_pos_set_batch_vector(vmn_pos* p, const uint8_t* e_be) {
    NONNULL(p);
    return p->batch_vector(e_be, p->e);
}
int vmn_pos_set_batch_vector_seed(vmn_pos* p, const uint8_t* seed, size_t seedlen) {
    NONNULL(p);
    return p->batch_vector_seed(seed, seedlen, p->e);
}
int vmn_pos_commit_prepare(vmn_pos* p) {
    NONNULL(p);
    return p->commit_prepare();
}
int vmn_posc_commit_prepare(vmn_posc* p) {
    NONNULL(p);
    return p->commit_prepare();
}
int vmn_ccpos_commit_prepare(vmn_ccpos* p) {
    NONNULL(p);
    return p->commit_prepare();
}
int vmn_pos_commit(vmn_pos* p, vmn_msg** commitment) {
    NONNULL(p);
    return p->commit(commitment);
}
int vmn_pos_reply(vmn_pos* p, const uint8_t* v_be, size_t vbytes, vmn_msg** reply) {
    NONNULL(p);
    return p->reply(v_be, vbytes, reply);
}
int vmn_pos_compute_af(vmn_pos* p) {
    NONNULL(p);
    return p->compute_af();
}
int vmn_pos_set_commitment(vmn_pos* p, const vmn_msg* commitment) {
    NONNULL(p);
    return p->set_commitment(commitment);
}
int vmn_pos_set_challenge(vmn_pos* p, const uint8_t* v_be, size_t vbytes) {
    NONNULL(p);
    return p->set_challenge(v_be, vbytes);
}
int vmn_pos_verify_prepare(vmn_pos* p, const vmn_msg* reply) {
    NONNULL(p);
    return p->verify_prepare(reply);
}
int vmn_pos_verify(vmn_pos* p, const vmn_msg* reply, int* verdict, int* verdicts5) {
    NONNULL(p);
    return p->verify(reply, verdict, verdicts5);
}
int vmn_pos_get_A(vmn_pos* p, uint8_t* out_be) {
    NONNULL(p);
    return p->get_A(out_be);
}
int vmn_pos_get_F(vmn_pos* p, uint8_t* out_be) {
    NONNULL(p);
    return p->get_F(out_be);
}
int vmn_pos_get_C(vmn_pos* p, uint8_t* out_be) {
    NONNULL(p);
    return p->get_CD(out_be, false);
}
int vmn_pos_get_D(vmn_pos* p, uint8_t* out_be) {
    NONNULL(p);
    return p->get_CD(out_be, true);
}
size_t vmn_pos_width(const vmn_pos* p) { return p ? p->width : 0; }

void vmn_posc_free(vmn_posc* p) {
    if (p) p->release_base_table();
    delete p;
}
int vmn_posc_set_instance(vmn_posc* p, const uint8_t* g_be, const vmn_garray* h, const vmn_garray* u, const vmn_rarray* r,
                          const uint32_t* pi) {
    NONNULL(p);
    return p->set_instance(g_be, h, u, r, pi);
}
int vmn_posc_set_batch_vector(vmn_posc* p, const uint8_t* e_be) {
    NONNULL(p);
    return p->batch_vector(e_be, p->e);
}
int vmn_posc_set_batch_vector_seed(vmn_posc* p, const uint8_t* seed, size_t seedlen) {
    NONNULL(p);
    return p->batch_vector_seed(seed, seedlen, p->e);
}
int vmn_posc_commit(vmn_posc* p, vmn_msg** commitment) {
    NONNULL(p);
    return p->commit(commitment);
}
int vmn_posc_reply(vmn_posc* p, const uint8_t* v_be, size_t vbytes, vmn_msg** reply) {
    NONNULL(p);
    return p->reply(v_be, vbytes, reply);
}
int vmn_posc_set_commitment(vmn_posc* p, const vmn_msg* commitment) {
    NONNULL(p);
    return p->set_commitment(commitment);
}
int vmn_posc_set_challenge(vmn_posc* p, const uint8_t* v_be, size_t vbytes) {
    NONNULL(p);
    return p->set_challenge(v_be, vbytes);
}
int vmn_posc_verify_prepare(vmn_posc* p, const vmn_msg* reply) {
    NONNULL(p);
    return p->verify_prepare(reply);
}
int vmn_posc_verify(vmn_posc* p, const vmn_msg* reply, int* verdict) {
    NONNULL(p);
    return p->verify(reply, verdict);
}
int vmn_posc_get_A(vmn_posc* p, uint8_t* out_be) {
    NONNULL(p);
    return p->get_ACD(out_be, 0);
}
int vmn_posc_get_C(vmn_posc* p, uint8_t* out_be) {
    NONNULL(p);
    return p->get_ACD(out_be, 1);
}
int vmn_posc_get_D(vmn_posc* p, uint8_t* out_be) {
    NONNULL(p);
    return p->get_ACD(out_be, 2);
}

void vmn_ccpos_free(vmn_ccpos* p) { delete p; }
int vmn_ccpos_set_instance(vmn_ccpos* p, const uint8_t* g_be, const vmn_garray* h, const vmn_garray* u, const uint8_t* pkey_be,
                           size_t width, const vmn_garray* const* w, const vmn_garray* const* wp, const vmn_rarray* r,
                           const uint32_t* pi, const vmn_rarray* const* s) {
    NONNULL(p);
    return p->set_instance(g_be, h, u, pkey_be, width, w, wp, r, pi, s);
}
int vmn_ccpos_set_batch_vector(vmn_ccpos* p, const uint8_t* e_be) {
    NONNULL(p);
    return p->batch_vector(e_be, p->e);
}
int vmn_ccpos_set_batch_vector_seed(vmn_ccpos* p, const uint8_t* seed, size_t seedlen) {
    NONNULL(p);
    return p->batch_vector_seed(seed, seedlen, p->e);
}
int vmn_ccpos_commit(vmn_ccpos* p, vmn_msg** commitment) {
    NONNULL(p);
    return p->commit(commitment);
}
int vmn_ccpos_reply(vmn_ccpos* p, const uint8_t* v_be, size_t vbytes, vmn_msg** reply) {
    NONNULL(p);
    return p->reply(v_be, vbytes, reply);
}
int vmn_ccpos_set_commitment(vmn_ccpos* p, const vmn_msg* commitment) {
    NONNULL(p);
    return p->set_commitment(commitment);
}
int vmn_ccpos_set_challenge(vmn_ccpos* p, const uint8_t* v_be, size_t vbytes) {
    NONNULL(p);
    return p->set_challenge(v_be, vbytes);
}
int vmn_ccpos_compute_ab(vmn_ccpos* p, const vmn_garray* raisedu) {
    NONNULL(p);
    return p->compute_ab(raisedu);
}
int vmn_ccpos_get_AB(vmn_ccpos* p, uint8_t* out_be, size_t* count) {
    NONNULL(p);
    return p->get_AB(out_be, count);
}
size_t vmn_ccpos_width(const vmn_ccpos* p) { return p ? p->width : 0; }
int vmn_ccpos_verify_prepare(vmn_ccpos* p, const vmn_msg* reply, const vmn_garray* raisedh, const uint8_t* rho_be, size_t rho_bytes) {
    NONNULL(p);
    return p->verify_prepare(reply, raisedh, rho_be, rho_bytes);
}
int vmn_ccpos_verify(vmn_ccpos* p, const vmn_msg* reply, const vmn_garray* raisedh, const uint8_t* rho_be, size_t rho_bytes,
                     int* verdict) {
    NONNULL(p);
    return p->verify(reply, raisedh, rho_be, rho_bytes, verdict);
}

// ---- threshold decryption --------------------------------------------------------------------------------------
int vmn_prod_factor(vmn_group* grp, int k, uint8_t* c_be) {
    if (!grp || !c_be || k < 1) return fail(VMN_ERR_ARG, "vmn_prod_factor: bad argument");
    HostGroup G;
    TRY(G.init(grp));
    Bytes b = G.ring_bytes(prod_factor(G, k));
    memcpy(c_be, b.data(), G.xb);
    return VMN_OK;
}
int vmn_lagrange_coefficients(vmn_group* grp, const uint8_t* correct, int k, int threshold, uint8_t* abs_be, int* negative) {
    if (!grp || !correct || !abs_be || !negative || k < 1 || threshold < 1) return fail(VMN_ERR_ARG, "vmn_lagrange_coefficients: bad argument");
    HostGroup G;
    TRY(G.init(grp));
    std::vector<Num> abs;
    std::vector<int> neg;
    TRY(lagrange(G, correct, k, threshold, abs, neg));
    for (size_t t = 0; t < abs.size(); ++t) {
        Bytes b = G.ring_bytes(abs[t]);
        memcpy(abs_be + t * G.xb, b.data(), G.xb);
        negative[t] = neg[t];
    }
    return VMN_OK;
}
int vmn_decryption_factors(vmn_group* grp, const vmn_garray* u, const uint8_t* secret_be, int k, vmn_garray** f_out) {
    if (!grp || !u || !secret_be || !f_out || k < 1) return fail(VMN_ERR_ARG, "vmn_decryption_factors: bad argument");
    HostGroup G;
    TRY(G.init(grp));
    // firstComponents.exp(secretKey.neg().mul(inverseFactor))   DistrElGamalSession.java:384-385
    Num ex = G.Zq.mul(G.Zq.neg(G.ring_from(secret_be)), G.Zq.inv(prod_factor(G, k)));
    Bytes eb = G.ring_bytes(ex);
    return vmn_garray_exp_scalar(u, eb.data(), eb.size(), f_out);
}
int vmn_combine_decryption_factors(vmn_group* grp, const vmn_garray* const* f, const uint8_t* correct, int k, int threshold,
                                   vmn_garray** out) {
    if (!grp || !f || !correct || !out) return fail(VMN_ERR_ARG, "vmn_combine_decryption_factors: null argument");
    HostGroup G;
    TRY(G.init(grp));
    std::vector<Num> abs;
    std::vector<int> neg, parties;
    TRY(lagrange(G, correct, k, threshold, abs, neg, &parties));
    GA pos, negp;                                   // products of the positive / negative parts (:465-503)
    for (size_t t = 0; t < parties.size(); ++t) {
        if (vmn::num64::is_zero(abs[t])) continue;
        const vmn_garray* base = f[parties[t]];
        if (!base) return fail(VMN_ERR_ARG, "vmn_combine_decryption_factors: factors of party %d are missing", parties[t]);
        Bytes eb = G.ring_bytes(abs[t]);
        GA tpow;
        TRY(vmn_garray_exp_scalar(base, eb.data(), eb.size(), tpow.out()));
        GA& acc = neg[t] ? negp : pos;
        if (!acc.p) {
            acc.p = tpow.release();
        } else {
            GA prod;
            TRY(vmn_garray_mul(acc, tpow, prod.out()));
            acc.reset();
            acc.p = prod.release();
        }
    }
    if (negp.p) {
        GA ninv;
        TRY(vmn_garray_inv(negp, ninv.out()));
        if (!pos.p) {
            *out = ninv.release();
            return VMN_OK;
        }
        return vmn_garray_mul(pos, ninv, out);
    }
    if (!pos.p) return fail(VMN_ERR_ARG, "vmn_combine_decryption_factors: all coefficients are zero");
    *out = pos.release();
    return VMN_OK;
}

int vmn_decproof_create(vmn_group* grp, int j, int k, int threshold, int ebitlen, const vmn_random_source* rs, vmn_decproof** out) {
    if (!grp || !out || k < 1 || j < 1 || j > k || threshold < 1 || threshold > k || ebitlen <= 0)
        return fail(VMN_ERR_ARG, "vmn_decproof_create: bad argument");
    std::unique_ptr<vmn_decproof> p(new vmn_decproof());
    TRY(p->init(grp, j, k, threshold, ebitlen, rs));
    *out = p.release();
    return VMN_OK;
}
void vmn_decproof_free(vmn_decproof* p) { delete p; }
int vmn_decproof_set_instance(vmn_decproof* p, const vmn_garray* u, const uint8_t* y_be, const vmn_garray* const* f) {
    NONNULL(p);
    return p->set_instance(u, y_be, f);
}
int vmn_decproof_set_batch_vector(vmn_decproof* p, const uint8_t* e_be) {
    NONNULL(p);
    if (!p->u || !e_be) return fail(VMN_ERR_ARG, "vmn_decproof_set_batch_vector: instance not set");
    return import_batch_vector(p->G.grp, e_be, vmn_garray_size(p->u), p->ebitlen, p->e);
}
int vmn_decproof_set_batch_vector_seed(vmn_decproof* p, const uint8_t* seed, size_t seedlen) {
    NONNULL(p);
    if (!p->u) return fail(VMN_ERR_ARG, "vmn_decproof_set_batch_vector_seed: instance not set");
    return vmn_rarray_from_prg(p->G.grp, seed, seedlen, vmn_garray_size(p->u), p->ebitlen, p->e.out());
}
int vmn_decproof_batch_input(vmn_decproof* p) {
    NONNULL(p);
    return p->batch_input();
}
int vmn_decproof_commit(vmn_decproof* p, const uint8_t* x_be, uint8_t* yp_out, uint8_t* Bp_out) {
    NONNULL(p);
    return p->commit(x_be, yp_out, Bp_out);
}
int vmn_decproof_reply(vmn_decproof* p, const uint8_t* v_be, size_t vbytes, uint8_t* kx_out) {
    NONNULL(p);
    return p->reply(v_be, vbytes, kx_out);
}
int vmn_decproof_set_commitment(vmn_decproof* p, int l, const uint8_t* yp_be, const uint8_t* Bp_be) {
    NONNULL(p);
    TRY(p->party(l));
    if (!yp_be || !Bp_be) return fail(VMN_ERR_ARG, "vmn_decproof_set_commitment: null argument");
    Bytes a(yp_be, yp_be + p->G.eb), b(Bp_be, Bp_be + p->G.eb);
    int ok = 1;
    TRY(p->G.check_elements({&a, &b}, &ok));
    if (!ok) return fail(VMN_ERR_FORMAT, "commitment holds a value that is not a group element");
    p->yp[l] = a;
    p->Bp[l] = b;
    return VMN_OK;
}
int vmn_decproof_set_reply(vmn_decproof* p, int l, const uint8_t* kx_be) {
    NONNULL(p);
    TRY(p->party(l));
    if (!kx_be) return fail(VMN_ERR_ARG, "vmn_decproof_set_reply: null argument");
    p->k_x[l] = p->G.ring_from(kx_be);
    p->have_kx[l] = 1;
    if (vmn::num64::cmp(p->k_x[l], p->G.Zq.n) >= 0) {          // pRing.toElement fails: k_x = 0, verdict false (:606-613)
        p->k_x[l] = Num(p->G.ql, 0);
        p->have_kx[l] = 2;
    }
    return VMN_OK;
}
int vmn_decproof_batch(vmn_decproof* p, int l) {
    NONNULL(p);
    TRY(p->party(l));
    if (!p->f[l] || !p->e.p) return fail(VMN_ERR_ARG, "vmn_decproof_batch: factors of party %d or the batching vector are missing", l);
    p->B[l].resize(p->G.eb);
    return vmn_garray_expprod(p->f[l], p->e, p->e_bits, p->B[l].data());              // :707-709
}
int vmn_decproof_verify(vmn_decproof* p, int l, const uint8_t* v_be, size_t vbytes, int* verdict) {
    NONNULL(p);
    return p->verify(l, v_be, vbytes, verdict);
}
int vmn_decproof_combine(vmn_decproof* p, const uint8_t* correct, const uint8_t* combinedy_be, const vmn_garray* combinedf) {
    NONNULL(p);
    return p->combine(correct, combinedy_be, combinedf);
}
int vmn_decproof_batch_combined(vmn_decproof* p) {
    NONNULL(p);
    return p->batch_combined();
}
int vmn_decproof_verify_combined(vmn_decproof* p, const uint8_t* v_be, size_t vbytes, int* verdict) {
    NONNULL(p);
    return p->verify_combined(v_be, vbytes, verdict);
}

// ---- independent generators, interactive --------------------------------------------------------------------------
int vmn_igen_create(vmn_group* grp, int j, int threshold, int ebitlen, const vmn_random_source* rs, vmn_igen** out) {
    if (!grp || !out || threshold < 1 || j < 1 || j > threshold || ebitlen <= 0) return fail(VMN_ERR_ARG, "vmn_igen_create: bad argument");
    std::unique_ptr<vmn_igen> p(new vmn_igen());
    TRY(p->init(grp, j, threshold, ebitlen, rs));
    *out = p.release();
    return VMN_OK;
}
void vmn_igen_free(vmn_igen* p) { delete p; }
int vmn_igen_set_instance(vmn_igen* p, const uint8_t* g_be, const vmn_garray* const* h, const vmn_rarray* s, const vmn_garray* combinedh) {
    NONNULL(p);
    if (!g_be || !h || !combinedh) return fail(VMN_ERR_ARG, "vmn_igen_set_instance: null argument");
    p->N = vmn_garray_size(combinedh);
    for (int l = 1; l <= p->threshold; ++l) {
        p->h[l] = h[l];
        if (h[l] && vmn_garray_size(h[l]) != p->N) return fail(VMN_ERR_ARG, "vmn_igen_set_instance: parts of party %d differ in size", l);
    }
    if (s && vmn_rarray_size(s) != p->N) return fail(VMN_ERR_ARG, "vmn_igen_set_instance: exponents differ in size");
    p->g.assign(g_be, g_be + p->G.eb);
    p->s = s;
    p->combinedh = combinedh;
    return VMN_OK;
}
int vmn_igen_set_batch_vector(vmn_igen* p, const uint8_t* e_be) {
    NONNULL(p);
    if (!p->combinedh || !e_be) return fail(VMN_ERR_ARG, "vmn_igen_set_batch_vector: instance not set");
    return import_batch_vector(p->G.grp, e_be, p->N, p->ebitlen, p->e);
}
int vmn_igen_set_batch_vector_seed(vmn_igen* p, const uint8_t* seed, size_t seedlen) {
    NONNULL(p);
    if (!p->combinedh) return fail(VMN_ERR_ARG, "vmn_igen_set_batch_vector_seed: instance not set");
    return vmn_rarray_from_prg(p->G.grp, seed, seedlen, p->N, p->ebitlen, p->e.out());          // :186-193
}
int vmn_igen_commit(vmn_igen* p, uint8_t* Ap_out) {
    NONNULL(p);
    return p->commit(Ap_out);
}
int vmn_igen_set_commitment(vmn_igen* p, int l, const uint8_t* Ap_be) {
    NONNULL(p);
    TRY(p->party(l));
    if (!Ap_be) return fail(VMN_ERR_ARG, "vmn_igen_set_commitment: null argument");
    Bytes a(Ap_be, Ap_be + p->G.eb);
    int ok = 1;
    TRY(p->G.check_elements({&a}, &ok));
    p->Ap[l] = ok ? a : p->G.one();                                               // :223-226
    return VMN_OK;
}
int vmn_igen_set_challenge(vmn_igen* p, const uint8_t* v_be, size_t vbytes) {
    NONNULL(p);
    if (!v_be || !vbytes) return fail(VMN_ERR_ARG, "vmn_igen_set_challenge: null argument");
    p->v = p->G.reduce(v_be, vbytes);
    return VMN_OK;
}
int vmn_igen_reply(vmn_igen* p, uint8_t* ka_out) {
    NONNULL(p);
    if (!ka_out || p->a.empty() || p->r.empty() || p->v.empty()) return fail(VMN_ERR_ARG, "vmn_igen_reply: needs commit() and the challenge");
    p->k_a[p->j] = p->G.mul_add(p->a, p->v, p->r);                                // k_a = a v + r   :246
    p->have[p->j] = 1;
    Bytes b = p->G.ring_bytes(p->k_a[p->j]);
    memcpy(ka_out, b.data(), p->G.xb);
    return VMN_OK;
}
int vmn_igen_set_reply(vmn_igen* p, int l, const uint8_t* ka_be) {
    NONNULL(p);
    TRY(p->party(l));
    if (!ka_be) return fail(VMN_ERR_ARG, "vmn_igen_set_reply: null argument");
    Num k = p->G.ring_from(ka_be);
    if (vmn::num64::cmp(k, p->G.Zq.n) >= 0) k = Num(p->G.ql, 0);                  // :263-266
    p->k_a[l] = k;
    p->have[l] = 1;
    return VMN_OK;
}
int vmn_igen_verify_combined(vmn_igen* p, int* verdict) {
    NONNULL(p);
    if (!verdict || !p->combinedh || !p->e.p || p->v.empty()) return fail(VMN_ERR_ARG, "vmn_igen_verify_combined: instance, batching vector or challenge missing");
    Num ksum(p->G.ql, 0);
    Bytes Aprod = p->G.one();
    for (int l = 1; l <= p->threshold; ++l) {
        if (p->Ap[l].empty() || !p->have[l]) return fail(VMN_ERR_ARG, "vmn_igen_verify_combined: commitment or reply of party %d missing", l);
        ksum = p->G.Zq.add(ksum, p->k_a[l]);
        Bytes t;
        TRY(p->G.el_mul(Aprod, p->Ap[l], t));
        Aprod = t;
    }
    return p->check(p->combinedh, Aprod, ksum, verdict);                          // :275-289
}
int vmn_igen_verify(vmn_igen* p, int l, int* verdict) {
    NONNULL(p);
    TRY(p->party(l));
    if (!verdict || !p->h[l] || !p->e.p || p->v.empty() || p->Ap[l].empty() || !p->have[l])
        return fail(VMN_ERR_ARG, "vmn_igen_verify: parts, batching vector, challenge, commitment or reply of party %d missing", l);
    return p->check(p->h[l], p->Ap[l], p->k_a[l], verdict);                       // :297-299
}

// ---- single elements ------------------------------------------------------------------------------------------------
namespace vmnp {
// one HostGroup per thread, rebuilt when the handle or the group behind it changes (a freed handle's address may be
// handed out again for another group: the cached modulus, order and kind are compared on every call)
const HostGroup* host_group(vmn_group* grp) {
    thread_local vmn_group* cached = nullptr;
    thread_local std::unique_ptr<HostGroup> hg;
    thread_local Bytes id;
    const size_t xb = vmn_group_exp_bytes(grp), eb = vmn_group_elem_bytes(grp);      // the modulus takes eb or eb / 2 bytes
    Bytes now(eb + xb + 1, 0);
    if (vmn_group_get_modulus(grp, now.data()) != VMN_OK || vmn_group_get_order(grp, now.data() + eb) != VMN_OK) return nullptr;
    now[eb + xb] = (uint8_t)vmn_group_kind(grp);
    if (cached != grp || !hg || id != now) {
        hg.reset(new HostGroup());
        if (hg->init(grp) != VMN_OK) {
            hg.reset();
            cached = nullptr;
            return nullptr;
        }
        cached = grp;
        id = now;
    }
    return hg.get();
}
}  // namespace vmnp
int vmn_element_exp(vmn_group* grp, const uint8_t* base_be, const uint8_t* e_be, size_t ebytes, uint8_t* out_be) {
    if (!grp || !base_be || !e_be || !ebytes || !out_be) return fail(VMN_ERR_ARG, "vmn_element_exp: null argument");
    const HostGroup* G = host_group(grp);
    if (!G) return VMN_ERR_ARG;
    Bytes out;
    TRY(G->el_exp(Bytes(base_be, base_be + G->eb), e_be, ebytes, out));
    memcpy(out_be, out.data(), G->eb);
    return VMN_OK;
}
int vmn_element_mul(vmn_group* grp, const uint8_t* a_be, const uint8_t* b_be, uint8_t* out_be) {
    if (!grp || !a_be || !b_be || !out_be) return fail(VMN_ERR_ARG, "vmn_element_mul: null argument");
    const HostGroup* G = host_group(grp);
    if (!G) return VMN_ERR_ARG;
    Bytes out;
    TRY(G->el_mul(Bytes(a_be, a_be + G->eb), Bytes(b_be, b_be + G->eb), out));
    memcpy(out_be, out.data(), G->eb);
    return VMN_OK;
}
int vmn_element_inv(vmn_group* grp, const uint8_t* a_be, uint8_t* out_be) {
    if (!grp || !a_be || !out_be) return fail(VMN_ERR_ARG, "vmn_element_inv: null argument");
    const HostGroup* G = host_group(grp);
    if (!G) return VMN_ERR_ARG;
    Bytes out;
    TRY(G->el_inv(Bytes(a_be, a_be + G->eb), out));
    memcpy(out_be, out.data(), G->eb);
    return VMN_OK;
}

}  // extern "C"
