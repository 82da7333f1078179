// vmnhip.hip — the C ABI of include/vmnhip.h over the gfx950 kernels of modp_kernels.h.
//
// Host side only orchestrates: it owns device buffers, picks window sizes, launches kernels on
// the context stream and does the O(1)-element sequential tails (hostbig.h).  There is no CPU
// implementation of any array operation in this library.
#include <stdarg.h>

#include <algorithm>
#include <memory>
#include <utility>

#include "modp_kernels.h"
#include "vmnhip_internal.h"

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
// (3072-bit needs 110 limbs: more modulus limbs than there are SGPRs; that size gets its own row
// generator and is not instantiated yet.)
#define VMN_FOR_SIZES(X) X(19, 16) X(37, 32) X(74, 64)

static bool size_for_bits(int nbits, int* S, int* NW) {
    const int sizes[][3] = {{512, 19, 16}, {1024, 37, 32}, {2048, 74, 64}};
    for (auto& s : sizes) {
        if (nbits <= s[0]) {
            *S = s[1];
            *NW = s[2];
            return true;
        }
    }
    return false;
}

static size_t lds_bytes(int S) { return (size_t)S * BLOCK * sizeof(u32); }
static int blocks_per_cu(int S) { return S <= 74 ? 2 : 1; }

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
        VMN_HIP(hipEventCreate(&rec.start));
        VMN_HIP(hipEventCreate(&rec.stop));
        VMN_HIP(hipEventRecord(rec.start, ctx->stream));
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(BLOCK), lds, ctx->stream, static_cast<KArgs>(args)...);
    VMN_HIP(hipGetLastError());
    if (ctx->timing) {
        VMN_HIP(hipEventRecord(rec.stop, ctx->stream));
        ctx->recs.push_back(rec);
    }
    return VMN_OK;
}

static int ensure_scratch(vmn_ctx* ctx, size_t bytes) {
    if (ctx->scratch_bytes >= bytes) return VMN_OK;
    if (ctx->scratch) {
        VMN_HIP(hipStreamSynchronize(ctx->stream));
        VMN_HIP(hipFree(ctx->scratch));
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
    }
    VMN_HIP(hipMalloc(&ctx->scratch, bytes));
    ctx->scratch_bytes = bytes;
    return VMN_OK;
}

// RAII device temporary on the context stream
struct DevTmp {
    vmn_ctx* ctx;
    void* p = nullptr;
    explicit DevTmp(vmn_ctx* c) : ctx(c) {}
    int alloc(size_t bytes) {
        VMN_HIP(hipMalloc(&p, bytes ? bytes : 16));
        return VMN_OK;
    }
    ~DevTmp() {
        if (p) {
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipFree(p);
        }
    }
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
    VMN_HIP(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;
    VMN_HIP(hipMalloc(&ctx->flags, 64 * sizeof(uint32_t)));
    VMN_HIP(hipMemsetAsync(ctx->flags, 0, 64 * sizeof(uint32_t), ctx->stream));
    *out = ctx.release();
    return VMN_OK;
}

extern "C" void vmn_ctx_destroy(vmn_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& r : ctx->recs) {
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->flags) (void)hipFree(ctx->flags);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

extern "C" int vmn_ctx_set_stream(vmn_ctx* ctx, void* hip_stream) {
    ARG_CHECK(ctx, "null ctx");
    VMN_HIP(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return VMN_OK;
}
extern "C" void* vmn_ctx_get_stream(vmn_ctx* ctx) { return ctx ? reinterpret_cast<void*>(ctx->stream) : nullptr; }
extern "C" int vmn_ctx_synchronize(vmn_ctx* ctx) {
    ARG_CHECK(ctx, "null ctx");
    VMN_HIP(hipStreamSynchronize(ctx->stream));
    return VMN_OK;
}
extern "C" int vmn_ctx_num_cus(vmn_ctx* ctx) { return ctx ? ctx->num_cus : 0; }

static int timing_collect(vmn_ctx* ctx) {
    if (ctx->recs.empty()) return VMN_OK;
    VMN_HIP(hipStreamSynchronize(ctx->stream));
    for (auto& r : ctx->recs) {
        float ms = 0;
        VMN_HIP(hipEventElapsedTime(&ms, r.start, r.stop));
        auto& acc = ctx->timing_acc[r.family];
        acc.first += 1;
        acc.second += ms;
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    ctx->recs.clear();
    return VMN_OK;
}
extern "C" int vmn_ctx_timing_enable(vmn_ctx* ctx, int on) {
    ARG_CHECK(ctx, "null ctx");
    VMN_TRY(timing_collect(ctx));
    ctx->timing = on != 0;
    return VMN_OK;
}
extern "C" int vmn_ctx_timing_reset(vmn_ctx* ctx) {
    ARG_CHECK(ctx, "null ctx");
    VMN_TRY(timing_collect(ctx));
    ctx->timing_acc.clear();
    return VMN_OK;
}
extern "C" int vmn_ctx_timing_get(vmn_ctx* ctx, const char* family, long* launches, double* total_ms) {
    ARG_CHECK(ctx && family, "null argument");
    VMN_TRY(timing_collect(ctx));
    auto it = ctx->timing_acc.find(family);
    if (launches) *launches = it == ctx->timing_acc.end() ? 0 : it->second.first;
    if (total_ms) *total_ms = it == ctx->timing_acc.end() ? 0.0 : it->second.second;
    return VMN_OK;
}

// ------------------------------------------------------------------------------------------------
// moduli and groups
// ------------------------------------------------------------------------------------------------
// words (NW, little-endian 32-bit) -> S limbs of 28 bits
static std::vector<uint32_t> words_to_limbs_host(const Big& w, int S) {
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

static int upload_words(vmn_ctx* ctx, uint32_t** dst, const std::vector<uint32_t>& v) {
    VMN_HIP(hipMalloc(dst, v.size() * sizeof(uint32_t)));
    VMN_HIP(hipMemcpy(*dst, v.data(), v.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    (void)ctx;
    return VMN_OK;
}

static void modulus_destroy(vmn_modulus& m) {
    if (m.d_n) (void)hipFree(m.d_n);
    if (m.d_rr) (void)hipFree(m.d_rr);
    if (m.d_one) (void)hipFree(m.d_one);
    delete m.hm;
    m = vmn_modulus();
}

// Set up one odd modulus given as big-endian bytes; S/NW are forced (q shares p's geometry).
static int modulus_init(vmn_ctx* ctx, vmn_modulus& m, const uint8_t* be, size_t nbytes, int S, int NW) {
    m.S = S;
    m.NW = NW;
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
    VMN_TRY(upload_words(ctx, &m.d_n, words_to_limbs_host(m.n_words, S)));
    VMN_TRY(upload_words(ctx, &m.d_one, words_to_limbs_host(r, S)));
    VMN_TRY(upload_words(ctx, &m.d_rr, words_to_limbs_host(rr, S)));
    m.hm = new hostbig::Mont(m.n_words);
    return VMN_OK;
}

extern "C" int vmn_modp_group_create(vmn_ctx* ctx, const uint8_t* p_be, const uint8_t* q_be, const uint8_t* g_be,
                                     size_t nbytes, vmn_group** out) {
    ARG_CHECK(ctx && p_be && q_be && g_be && out && nbytes > 0, "null argument");
    VMN_HIP(hipSetDevice(ctx->device));
    Big pw = hostbig::from_be(p_be, nbytes, (nbytes + 3) / 4);
    int nbits = hostbig::bit_length(pw);
    int S, NW;
    if (!size_for_bits(nbits, &S, &NW)) {
        set_error("vmn_modp_group_create: %d-bit modulus not supported (max 2048)", nbits);
        return VMN_ERR_UNSUPPORTED;
    }
    std::unique_ptr<vmn_group> g(new vmn_group());
    g->ctx = ctx;
    g->nbytes = nbytes;
    int rc = modulus_init(ctx, g->P, p_be, nbytes, S, NW);
    if (rc == VMN_OK) rc = modulus_init(ctx, g->Q, q_be, nbytes, S, NW);
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
    (void)hipStreamSynchronize(grp->ctx->stream);
    for (auto& kv : grp->fixed) {
        if (kv.second.d_tab) (void)hipFree(kv.second.d_tab);
    }
    modulus_destroy(grp->P);
    modulus_destroy(grp->Q);
    delete grp;
}
extern "C" size_t vmn_group_elem_bytes(const vmn_group* grp) { return grp ? grp->nbytes : 0; }
extern "C" size_t vmn_group_exp_bytes(const vmn_group* grp) { return grp ? grp->nbytes : 0; }

// ------------------------------------------------------------------------------------------------
// generic array plumbing (group arrays are residues mod p, ring arrays residues mod q)
// ------------------------------------------------------------------------------------------------
static size_t elem_words(const vmn_modulus& m) { return (size_t)stride_for_limbs(m.S); }
static unsigned grid_for(size_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK); }

static int alloc_elems(const vmn_modulus& m, size_t n, uint32_t** d) {
    VMN_HIP(hipMalloc(d, std::max<size_t>(n, 1) * elem_words(m) * sizeof(uint32_t)));
    return VMN_OK;
}

static int import_be(vmn_ctx* ctx, const vmn_modulus& m, size_t nbytes, const uint8_t* be, size_t n, uint32_t* d_out,
                     int* all_in_range) {
    if (all_in_range) *all_in_range = 1;
    if (n == 0) return VMN_OK;
    DevTmp raw(ctx);
    VMN_TRY(raw.alloc(n * nbytes + 8));
    VMN_HIP(hipMemcpyAsync(raw.p, be, n * nbytes, hipMemcpyHostToDevice, ctx->stream));
    VMN_HIP(hipMemsetAsync(ctx->flags, 0, sizeof(uint32_t), ctx->stream));
    int rc = VMN_ERR_ARG;
#define X(S_, NW_)                                                                                              \
    if (m.S == S_)                                                                                              \
        rc = launch(ctx, "import", k_import_be<S_, NW_>, grid_for(n), lds_bytes(S_), d_out, raw.as<uint8_t>(), \
                    nbytes, n, m.d_n, m.n0inv, m.d_rr, ctx->flags);
    VMN_FOR_SIZES(X)
#undef X
    VMN_TRY(rc);
    uint32_t fl = 0;
    VMN_HIP(hipMemcpyAsync(&fl, ctx->flags, sizeof(fl), hipMemcpyDeviceToHost, ctx->stream));
    VMN_HIP(hipStreamSynchronize(ctx->stream));
    if (all_in_range) *all_in_range = (fl & 1u) ? 0 : 1;
    return VMN_OK;
}

static int export_be(vmn_ctx* ctx, const vmn_modulus& m, size_t nbytes, const uint32_t* d_in, size_t n, uint8_t* be) {
    if (n == 0) return VMN_OK;
    DevTmp raw(ctx);
    VMN_TRY(raw.alloc(n * nbytes + 8));
    int rc = VMN_ERR_ARG;
#define X(S_, NW_)                                                                                          \
    if (m.S == S_)                                                                                          \
        rc = launch(ctx, "export", k_export_be<S_, NW_>, grid_for(n), lds_bytes(S_), raw.as<uint8_t>(),    \
                    nbytes, d_in, n, m.d_n, m.n0inv);
    VMN_FOR_SIZES(X)
#undef X
    VMN_TRY(rc);
    VMN_HIP(hipMemcpyAsync(be, raw.p, n * nbytes, hipMemcpyDeviceToHost, ctx->stream));
    VMN_HIP(hipStreamSynchronize(ctx->stream));
    return VMN_OK;
}

// one element (big-endian) -> device, M28 form
static int import_one(vmn_ctx* ctx, const vmn_modulus& m, size_t nbytes, const uint8_t* be, uint32_t** d_out) {
    VMN_TRY(alloc_elems(m, 1, d_out));
    int ok = 1;
    int rc = import_be(ctx, m, nbytes, be, 1, *d_out, &ok);
    if (rc != VMN_OK) {
        (void)hipFree(*d_out);
        *d_out = nullptr;
        return rc;
    }
    if (!ok) {
        (void)hipFree(*d_out);
        *d_out = nullptr;
        set_error("scalar operand out of range");
        return VMN_ERR_FORMAT;
    }
    return VMN_OK;
}

static int mul_arrays(vmn_ctx* ctx, const vmn_modulus& m, const uint32_t* x, const uint32_t* y, size_t ystride, size_t n,
                      uint32_t* out) {
    if (n == 0) return VMN_OK;
    int rc = VMN_ERR_ARG;
#define X(S_, NW_) \
    if (m.S == S_) rc = launch(ctx, "modmul", k_mul<S_>, grid_for(n), lds_bytes(S_), out, x, y, ystride, n, m.d_n, m.n0inv);
    VMN_FOR_SIZES(X)
#undef X
    return rc;
}

// M28 residues -> packed words (n * NW words)
static int to_words(vmn_ctx* ctx, const vmn_modulus& m, const uint32_t* in, size_t n, uint32_t* out_words) {
    if (n == 0) return VMN_OK;
    int rc = VMN_ERR_ARG;
#define X(S_, NW_) \
    if (m.S == S_) rc = launch(ctx, "to_words", k_to_words<S_, NW_>, grid_for(n), lds_bytes(S_), out_words, in, n, m.d_n, m.n0inv);
    VMN_FOR_SIZES(X)
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
static int modpow_words(vmn_ctx* ctx, const vmn_modulus& m, const uint32_t* x, const uint32_t* e_words, int ewords,
                        size_t estride, int ebits, size_t n, uint32_t* out) {
    if (n == 0) return VMN_OK;
    if (ebits < 1) ebits = 1;
    int wbits = pick_window(ebits);
    unsigned max_blocks = (unsigned)(ctx->num_cus * blocks_per_cu(m.S));
    unsigned grid = std::min<unsigned>(grid_for(n), max_blocks);
    size_t tab_bytes = (size_t)grid * BLOCK * ((size_t)1 << wbits) * elem_words(m) * sizeof(uint32_t);
    VMN_TRY(ensure_scratch(ctx, tab_bytes));
    int rc = VMN_ERR_ARG;
#define X(S_, NW_)                                                                                                 \
    if (m.S == S_)                                                                                                 \
        rc = launch(ctx, "modpow", k_modpow<S_>, grid, lds_bytes(S_), out, x, e_words, ewords, estride, ebits, wbits, \
                    n, m.d_n, m.n0inv, m.d_one, reinterpret_cast<uint32_t*>(ctx->scratch));
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
    VMN_TRY(alloc_elems(grp->P, n, &a->d));
    *out = a.release();
    return VMN_OK;
}
static int new_rarray(vmn_group* grp, size_t n, vmn_rarray** out) {
    std::unique_ptr<vmn_rarray> a(new vmn_rarray());
    a->grp = grp;
    a->n = n;
    VMN_TRY(alloc_elems(grp->Q, n, &a->d));
    *out = a.release();
    return VMN_OK;
}

extern "C" void vmn_garray_free(vmn_garray* a) {
    if (!a) return;
    (void)hipStreamSynchronize(a->grp->ctx->stream);
    if (a->d) (void)hipFree(a->d);
    delete a;
}
extern "C" void vmn_rarray_free(vmn_rarray* a) {
    if (!a) return;
    (void)hipStreamSynchronize(a->grp->ctx->stream);
    if (a->d) (void)hipFree(a->d);
    delete a;
}
extern "C" size_t vmn_garray_size(const vmn_garray* a) { return a ? a->n : 0; }
extern "C" size_t vmn_rarray_size(const vmn_rarray* a) { return a ? a->n : 0; }

extern "C" int vmn_garray_from_be(vmn_group* grp, const uint8_t* be, size_t n, vmn_garray** out, int* all_in_range) {
    ARG_CHECK(grp && out && (be || n == 0), "null argument");
    VMN_HIP(hipSetDevice(grp->ctx->device));
    vmn_garray* a = nullptr;
    VMN_TRY(new_garray(grp, n, &a));
    int rc = import_be(grp->ctx, grp->P, grp->nbytes, be, n, a->d, all_in_range);
    if (rc != VMN_OK) {
        vmn_garray_free(a);
        return rc;
    }
    *out = a;
    return VMN_OK;
}
extern "C" int vmn_rarray_from_be(vmn_group* grp, const uint8_t* be, size_t n, vmn_rarray** out, int* all_in_range) {
    ARG_CHECK(grp && out && (be || n == 0), "null argument");
    VMN_HIP(hipSetDevice(grp->ctx->device));
    vmn_rarray* a = nullptr;
    VMN_TRY(new_rarray(grp, n, &a));
    int rc = import_be(grp->ctx, grp->Q, grp->nbytes, be, n, a->d, all_in_range);
    if (rc != VMN_OK) {
        vmn_rarray_free(a);
        return rc;
    }
    *out = a;
    return VMN_OK;
}
extern "C" int vmn_garray_to_be(const vmn_garray* a, uint8_t* be_out) {
    ARG_CHECK(a && (be_out || a->n == 0), "null argument");
    VMN_HIP(hipSetDevice(a->grp->ctx->device));
    return export_be(a->grp->ctx, a->grp->P, a->grp->nbytes, a->d, a->n, be_out);
}
extern "C" int vmn_rarray_to_be(const vmn_rarray* a, uint8_t* be_out) {
    ARG_CHECK(a && (be_out || a->n == 0), "null argument");
    VMN_HIP(hipSetDevice(a->grp->ctx->device));
    return export_be(a->grp->ctx, a->grp->Q, a->grp->nbytes, a->d, a->n, be_out);
}

extern "C" int vmn_garray_mul(const vmn_garray* x, const vmn_garray* y, vmn_garray** out) {
    ARG_CHECK(x && y && out, "null argument");
    ARG_CHECK(x->grp == y->grp && x->n == y->n, "arrays differ in group or size");
    vmn_group* g = x->grp;
    VMN_HIP(hipSetDevice(g->ctx->device));
    vmn_garray* r = nullptr;
    VMN_TRY(new_garray(g, x->n, &r));
    int rc = mul_arrays(g->ctx, g->P, x->d, y->d, elem_words(g->P), x->n, r->d);
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
    vmn_ctx* ctx = g->ctx;
    VMN_HIP(hipSetDevice(ctx->device));
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
    vmn_ctx* ctx = g->ctx;
    VMN_HIP(hipSetDevice(ctx->device));
    if (ebits <= 0 || (size_t)ebits > 8 * ebytes) ebits = (int)(8 * ebytes);
    int ewords = (ebits + 31) / 32;
    std::vector<uint32_t> hw;
    be_ints_to_words(exps_be, ebytes, x->n, ewords, hw);
    vmn_garray* r = nullptr;
    VMN_TRY(new_garray(g, x->n, &r));
    DevTmp ew(ctx);
    int rc = ew.alloc(hw.size() * sizeof(uint32_t));
    if (rc == VMN_OK && !hw.empty()) {
        hipError_t he = hipMemcpyAsync(ew.p, hw.data(), hw.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(ctx->stream);   // hw is pageable host memory
        if (he != hipSuccess) {
            set_error("exponent upload failed: %s", hipGetErrorString(he));
            rc = VMN_ERR_DEVICE;
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
    vmn_ctx* ctx = g->ctx;
    VMN_HIP(hipSetDevice(ctx->device));
    int ewords = (int)((ebytes + 3) / 4);
    Big e = hostbig::from_be(e_be, ebytes, ewords);
    int ebits = std::max(1, hostbig::bit_length(e));
    ewords = (ebits + 31) / 32;
    vmn_garray* r = nullptr;
    VMN_TRY(new_garray(g, x->n, &r));
    DevTmp ew(ctx);
    int rc = ew.alloc(ewords * sizeof(uint32_t));
    if (rc == VMN_OK) {
        hipError_t he = hipMemcpy(ew.p, e.data(), ewords * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (he != hipSuccess) {
            set_error("exponent upload failed: %s", hipGetErrorString(he));
            rc = VMN_ERR_DEVICE;
        }
    }
    if (rc == VMN_OK) rc = modpow_words(ctx, g->P, x->d, ew.as<uint32_t>(), ewords, 0, ebits, x->n, r->d);
    if (rc != VMN_OK) {
        vmn_garray_free(r);
        return rc;
    }
    *out = r;
    return VMN_OK;
}
