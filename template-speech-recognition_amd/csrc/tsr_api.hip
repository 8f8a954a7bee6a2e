// tsr_api.hip -- the C ABI of libtsr_hip.so (declared in include/tsr.h).
// Host-side validation, device buffer ownership, and kernel sequencing.  All
// arithmetic lives in detect_exact.hip / detect_mfma.hip / parts.hip.
#include <dlfcn.h>
#include <math.h>
#include <cmath>
#include <stdarg.h>

#include <algorithm>
#include <new>

#include "tsr_internal.h"

#include <mutex>
#include <unordered_map>

// ---- errors ---------------------------------------------------------------------
static thread_local char g_err[512] = "";

void tsr_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

TSR_API int tsr_version(void) { return TSR_VERSION; }

TSR_API const char *tsr_last_error(void) { return g_err; }

TSR_API const char *tsr_strerror(int status)
{
    switch (status) {
    case TSR_OK: return "ok";
    case TSR_ERR_INVALID: return "invalid argument";
    case TSR_ERR_HIP: return "HIP runtime error";
    case TSR_ERR_NO_DEVICE: return "no usable GPU device";
    case TSR_ERR_UNSUPPORTED: return "unsupported request";
    case TSR_ERR_NONFINITE: return "non-finite filter value";
    case TSR_ERR_RCCL: return "RCCL error";
    default: return "unknown status";
    }
}

// ---- context ---------------------------------------------------------------------
TSR_API int tsr_device_count(int *count)
{
    TSR_REQUIRE(count, "tsr_device_count: null pointer");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return TSR_OK;
}

TSR_API int tsr_ctx_destroy(tsr_ctx *ctx);

TSR_API int tsr_ctx_create(int device, tsr_ctx **out)
{
    TSR_REQUIRE(out, "tsr_ctx_create: null out pointer");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        tsr_set_error("tsr_ctx_create: no HIP device is visible (this library has no CPU path)");
        return TSR_ERR_NO_DEVICE;
    }
    TSR_REQUIRE(device >= 0 && device < n, "tsr_ctx_create: device %d out of range [0,%d)", device, n);
    TSR_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    TSR_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        tsr_set_error("tsr_ctx_create: device %d is %s; this library is built for gfx950 only", device,
                      prop.gcnArchName);
        return TSR_ERR_NO_DEVICE;
    }
    tsr_ctx *ctx = new (std::nothrow) tsr_ctx;
    TSR_REQUIRE(ctx, "tsr_ctx_create: out of host memory");
    ctx->device = device;
    ctx->cus = prop.multiProcessorCount;
    ctx->hbm_bytes = (int64_t)prop.totalGlobalMem;
    snprintf(ctx->name, sizeof ctx->name, "%s (%s)", prop.name, prop.gcnArchName);
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev0);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev1);
    if (e != hipSuccess) {
        tsr_set_error("tsr_ctx_create: stream/event creation failed: %s", hipGetErrorString(e));
        tsr_ctx_destroy(ctx);  // frees whatever was created
        return TSR_ERR_HIP;
    }
    // option defaults: the environment is consulted here and nowhere else
    if (const char *e = getenv("TSR_FASTK")) ctx->opt_fast_kernel = e[0] == 't' ? TSR_FAST_KERNEL_TAP2 : (e[0] == 'o' ? TSR_FAST_KERNEL_INTERLEAVED : TSR_OPT_AUTO);
    if (const char *e = getenv("TSR_EXACTK")) ctx->opt_exact_kernel = e[0] == 'v' ? TSR_EXACT_KERNEL_VALU : (e[0] == 'm' ? TSR_EXACT_KERNEL_MFMA : TSR_OPT_AUTO);
    if (const char *e = getenv("TSR_PART_LAYOUT")) ctx->opt_part_layout = !strcmp(e, "bytes") ? TSR_PART_LAYOUT_BYTES : (!strcmp(e, "bits") ? TSR_PART_LAYOUT_BITS : TSR_OPT_AUTO);
    if (const char *e = getenv("TSR_XCD_MAP")) ctx->opt_xcd_map = e[0] != '0';
    *out = ctx;
    return TSR_OK;
}

TSR_API int tsr_ctx_set_option(tsr_ctx *ctx, int option, int value)
{
    TSR_REQUIRE(ctx, "tsr_ctx_set_option: null ctx");
    switch (option) {
    case TSR_OPT_FAST_KERNEL:
        TSR_REQUIRE(value >= 0 && value <= TSR_FAST_KERNEL_INTERLEAVED, "tsr_ctx_set_option: bad value %d for TSR_OPT_FAST_KERNEL", value);
        ctx->opt_fast_kernel = value;
        return TSR_OK;
    case TSR_OPT_EXACT_KERNEL:
        TSR_REQUIRE(value >= 0 && value <= TSR_EXACT_KERNEL_MFMA, "tsr_ctx_set_option: bad value %d for TSR_OPT_EXACT_KERNEL", value);
        ctx->opt_exact_kernel = value;
        return TSR_OK;
    case TSR_OPT_PART_LAYOUT:
        TSR_REQUIRE(value >= 0 && value <= TSR_PART_LAYOUT_BITS, "tsr_ctx_set_option: bad value %d for TSR_OPT_PART_LAYOUT", value);
        ctx->opt_part_layout = value;
        return TSR_OK;
    case TSR_OPT_XCD_MAP:
        TSR_REQUIRE(value == 0 || value == 1, "tsr_ctx_set_option: bad value %d for TSR_OPT_XCD_MAP", value);
        ctx->opt_xcd_map = value;
        return TSR_OK;
    default:
        break;
    }
    tsr_set_error("tsr_ctx_set_option: unknown option %d", option);
    return TSR_ERR_INVALID;
}

TSR_API int tsr_ctx_get_option(const tsr_ctx *ctx, int option, int *value)
{
    TSR_REQUIRE(ctx && value, "tsr_ctx_get_option: null pointer");
    switch (option) {
    case TSR_OPT_FAST_KERNEL: *value = ctx->opt_fast_kernel; return TSR_OK;
    case TSR_OPT_EXACT_KERNEL: *value = ctx->opt_exact_kernel; return TSR_OK;
    case TSR_OPT_PART_LAYOUT: *value = ctx->opt_part_layout; return TSR_OK;
    case TSR_OPT_XCD_MAP: *value = ctx->opt_xcd_map; return TSR_OK;
    case TSR_INFO_EXACT_MFMA:  // 1 trusted, -1 probe ran and differed, -2 last probe could not run (it is tried again), 0 not probed
        *value = ctx->exact_mfma_state != 0 ? ctx->exact_mfma_state : (ctx->exact_mfma_could_not_run ? -2 : 0);
        return TSR_OK;
    case TSR_INFO_LAST_BANK_KERNEL: *value = ctx->last_bank_kernel; return TSR_OK;
    default: break;
    }
    tsr_set_error("tsr_ctx_get_option: unknown option %d", option);
    return TSR_ERR_INVALID;
}

TSR_API int tsr_ctx_destroy(tsr_ctx *ctx)
{
    if (!ctx) return TSR_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm) tsr_dist_finalize(ctx);
    for (auto &b : ctx->scratch)
        if (b.ptr) (void)hipFree(b.ptr);
    if (ctx->pinned.ptr) (void)hipHostFree(ctx->pinned.ptr);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return TSR_OK;
}

TSR_API int tsr_ctx_sync(tsr_ctx *ctx)
{
    TSR_REQUIRE(ctx, "tsr_ctx_sync: null ctx");
    TSR_HIP(hipStreamSynchronize(ctx->stream));
    return TSR_OK;
}

TSR_API int tsr_timer_start(tsr_ctx *ctx)
{
    TSR_REQUIRE(ctx, "tsr_timer_start: null ctx");
    TSR_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    return TSR_OK;
}

TSR_API int tsr_timer_stop(tsr_ctx *ctx, float *elapsed_ms)
{
    TSR_REQUIRE(ctx && elapsed_ms, "tsr_timer_stop: null pointer");
    TSR_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    TSR_HIP(hipEventSynchronize(ctx->ev1));
    TSR_HIP(hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
    return TSR_OK;
}

TSR_API int tsr_device_info(tsr_ctx *ctx, char *name, int name_len, int *cus, int64_t *hbm_bytes)
{
    TSR_REQUIRE(ctx, "tsr_device_info: null ctx");
    if (name && name_len > 0) snprintf(name, (size_t)name_len, "%s", ctx->name);
    if (cus) *cus = ctx->cus;
    if (hbm_bytes) *hbm_bytes = ctx->hbm_bytes;
    return TSR_OK;
}

int tsr_scratch(tsr_ctx *ctx, int slot, size_t bytes, void **out)
{
    tsr_devbuf &b = ctx->scratch[slot];
    if (bytes > b.cap) {
        if (b.ptr) {
            TSR_HIP(hipStreamSynchronize(ctx->stream));
            TSR_HIP(hipFree(b.ptr));
            b.ptr = nullptr;
            b.cap = 0;
        }
        size_t cap = std::max<size_t>(bytes, 1 << 20);
        TSR_HIP(hipMalloc(&b.ptr, cap));
        b.cap = cap;
    }
    *out = b.ptr;
    return TSR_OK;
}

int tsr_download(tsr_ctx *ctx, void *dst, const void *d_src, size_t bytes)
{
    if (bytes == 0) return TSR_OK;
    if (bytes < ((size_t)256 << 10) || bytes > ((size_t)256 << 20)) {
        TSR_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        TSR_HIP(hipStreamSynchronize(ctx->stream));
        return TSR_OK;
    }
    tsr_devbuf &b = ctx->pinned;
    if (bytes > b.cap) {
        if (b.ptr) TSR_HIP(hipHostFree(b.ptr));
        b.ptr = nullptr;
        b.cap = 0;
        TSR_HIP(hipHostMalloc(&b.ptr, bytes, hipHostMallocDefault));
        b.cap = bytes;
    }
    TSR_HIP(hipMemcpyAsync(b.ptr, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    TSR_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(dst, b.ptr, bytes);
    return TSR_OK;
}

namespace {

template <typename T>
int dev_upload(tsr_ctx *ctx, T **dptr, const T *src, size_t count, size_t slack_bytes = 0)
{
    *dptr = nullptr;
    size_t bytes = count * sizeof(T);
    TSR_HIP(hipMalloc(reinterpret_cast<void **>(dptr), std::max<size_t>(bytes + slack_bytes, 16)));
    if (slack_bytes) TSR_HIP(hipMemsetAsync(reinterpret_cast<char *>(*dptr) + bytes, 0, slack_bytes, ctx->stream));
    if (bytes) TSR_HIP(hipMemcpyAsync(*dptr, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return TSR_OK;
}

void dev_free(void *p)
{
    if (p) (void)hipFree(p);
}

// ---- a small pool for the large arrays of corpora -------------------------------------------------------------
// tsr_corpus_part_features / tsr_edge_corpus_create make a new corpus per call; its feature form, result arrays and
// utterance tables are hipMalloc'ed, and a corpus that goes away hipFree's them (each a device-wide synchronisation
// and, for the next call, a fresh allocation: ~0.2-0.4 ms per call at config 3, as much as the kernels it feeds).
// Blocks of destroyed corpora are kept here instead (process-wide: a corpus may outlive its context) and handed to the
// next corpus that asks for about the same size.  Bounded (kPoolMaxBytes); tsr_pool_trim() gives everything back.
namespace {
struct PoolBlock {
    void *p;
    size_t bytes;
    int device;
};
struct DevPool {
    std::mutex mu;
    std::vector<PoolBlock> idle;
    std::unordered_map<void *, std::pair<size_t, int>> live;  // blocks handed out: size, device
    size_t held = 0;
};
DevPool &pool()
{
    static DevPool *g = new DevPool;  // (never destroyed: no teardown-order problems at process exit)
    return *g;
}
constexpr size_t kPoolMaxBytes = (size_t)2 << 30, kPoolMinBlock = (size_t)64 << 10;

hipError_t pool_malloc(int device, void **out, size_t bytes)
{
    bytes = std::max<size_t>(bytes, 16);
    DevPool &pl = pool();
    if (bytes >= kPoolMinBlock) {
        std::lock_guard<std::mutex> g(pl.mu);
        size_t best = (size_t)-1;
        for (size_t i = 0; i < pl.idle.size(); ++i) {
            const PoolBlock &b = pl.idle[i];
            if (b.device == device && b.bytes >= bytes && b.bytes <= bytes + bytes / 4 && (best == (size_t)-1 || b.bytes < pl.idle[best].bytes)) best = i;
        }
        if (best != (size_t)-1) {
            const PoolBlock b = pl.idle[best];
            pl.idle.erase(pl.idle.begin() + (long)best);
            pl.held -= b.bytes;
            pl.live[b.p] = {b.bytes, b.device};
            *out = b.p;
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess) {  // give the pool's idle blocks back and try once more
        tsr_pool_trim();
        e = hipMalloc(out, bytes);
    }
    if (e == hipSuccess && bytes >= kPoolMinBlock) {
        std::lock_guard<std::mutex> g(pl.mu);
        pl.live[*out] = {bytes, device};
    }
    return e;
}

void pool_free(void *p)  // (the caller has synchronised the device: nothing in flight uses the block)
{
    if (!p) return;
    DevPool &pl = pool();
    {
        std::lock_guard<std::mutex> g(pl.mu);
        auto it = pl.live.find(p);
        if (it != pl.live.end()) {
            const size_t bytes = it->second.first;
            const int device = it->second.second;
            pl.live.erase(it);
            if (pl.held + bytes <= kPoolMaxBytes) {
                pl.idle.push_back({p, bytes, device});
                pl.held += bytes;
                return;
            }
        }
    }
    (void)hipFree(p);
}
}  // namespace

TSR_API int tsr_pool_trim(void)
{
    DevPool &pl = pool();
    std::vector<PoolBlock> idle;
    {
        std::lock_guard<std::mutex> g(pl.mu);
        idle.swap(pl.idle);
        pl.held = 0;
    }
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (const PoolBlock &b : idle) {
        (void)hipSetDevice(b.device);
        (void)hipFree(b.p);
    }
    (void)hipSetDevice(cur);
    return TSR_OK;
}

enum { SLOT_RAW = 0, SLOT_IN = 1, SLOT_W = 2, SLOT_OUT = 3, SLOT_AUX = 4, SLOT_AUX2 = 5, SLOT_LIST = 6, SLOT_CNT = 7 };

}  // namespace

// ---- bank ----------------------------------------------------------------------
static int bank_build_exact(tsr_ctx *ctx, tsr_bank *b)
{
    // Wx[g][tau][d][m]  (see tsr_internal.h)
    const int M = b->M;
    const int64_t D = b->D;
    const int WR = (int)((D + 31) / 32);
    b->Dq = WR;
    b->n_groups = (M + TSR_MG - 1) / TSR_MG;
    b->gx_off.assign(b->n_groups, 0);
    b->gx_L.assign(b->n_groups, 0);
    int64_t total = 0;
    for (int g = 0; g < b->n_groups; ++g) {
        int Lg = 0;
        for (int m = g * TSR_MG; m < std::min(M, (g + 1) * TSR_MG); ++m) Lg = std::max(Lg, b->L[m]);
        b->gx_off[g] = total;
        b->gx_L[g] = Lg;
        total += (int64_t)Lg * WR * 32 * TSR_MG;
    }
    std::vector<float> Wx((size_t)total, 0.0f);
    int64_t woff = 0;
    for (int m = 0; m < M; ++m) {
        const int g = m / TSR_MG, mi = m % TSR_MG;
        float *dst = Wx.data() + b->gx_off[g];
        const float *src = b->W_cat.data() + woff;
        for (int tau = 0; tau < b->L[m]; ++tau)
            for (int64_t d = 0; d < D; ++d)
                dst[((int64_t)tau * WR * 32 + d) * TSR_MG + mi] = src[(int64_t)tau * D + d];
        woff += (int64_t)b->L[m] * D;
    }
    std::vector<float> cf(M);
    for (int m = 0; m < M; ++m) cf[m] = (float)b->c[m];
    // matrix-core form of the same chains (detect_exact_mfma.hip): column j = (template j & 7, replica j >> 3) of
    // K-step (tau', features 4 n .. 4 n + 3) holds W_m[tau' - replica][4 n + k]; lane = 16 k + j, four K-steps per
    // float4.  Only for banks whose sums can never be subnormal and whose weights are finite (the matrix cores are
    // not trusted with either): every non-zero |w| >= 2^-100 makes every partial sum a multiple of 2^-123.
    bool mfma_ok = true;
    for (float w : b->W_cat) mfma_ok = mfma_ok && std::isfinite(w) && (w == 0.0f || std::fabs(w) >= 0x1p-100f);
    const int Dp = (int)((D + 15) / 16 * 16), NK4 = Dp / 16;
    std::vector<int64_t> gm_off(b->n_groups, 0);
    std::vector<int32_t> gm_rows(b->n_groups, 0);
    std::vector<float> Wm;
    if (mfma_ok) {
        int64_t total4 = 0;
        for (int g = 0; g < b->n_groups; ++g) {
            // the kernel takes two groups at a time through the taps: both get the pair's number of tap rows
            const int mate = (g ^ 1) < b->n_groups ? (g ^ 1) : g;
            gm_rows[g] = std::max(b->gx_L[g], b->gx_L[mate]) + 1;
            gm_off[g] = total4;
            total4 += (int64_t)gm_rows[g] * NK4 * 64;
        }
        Wm.assign((size_t)(total4 + 128) * 4, 0.0f);  // (+ two fragments: the kernel loads up to two steps ahead)
        std::vector<int64_t> wstart(M + 1, 0);
        for (int m = 0; m < M; ++m) wstart[m + 1] = wstart[m] + (int64_t)b->L[m] * D;
        for (int g = 0; g < b->n_groups; ++g)
            for (int tp = 0; tp < gm_rows[g]; ++tp)
                for (int n4 = 0; n4 < NK4; ++n4)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int k = lane >> 4, j = lane & 15, m = g * TSR_MG + (j & 7), tau = tp - (j >> 3);
                        if (m >= M || tau < 0 || tau >= b->L[m]) continue;
                        float *dst = Wm.data() + (size_t)(gm_off[g] + ((int64_t)tp * NK4 + n4) * 64 + lane) * 4;
                        for (int u = 0; u < 4; ++u) {
                            const int64_t d = 16 * n4 + 4 * u + k;
                            if (d < D) dst[u] = b->W_cat[(size_t)(wstart[m] + (int64_t)tau * D + d)];
                        }
                    }
    }
    // new device forms first; the bank's pointers change only when all of them exist, so a failed
    // (re)build leaves the old, self-consistent bank behind
    float *n_Wx = nullptr, *n_cf = nullptr, *n_Wm = nullptr;
    int64_t *n_off = nullptr, *n_gm = nullptr;
    int32_t *n_gr = nullptr;
    int32_t *n_gL = nullptr, *n_L = nullptr;
    int s = dev_upload(ctx, &n_Wx, Wx.data(), Wx.size());
    if (s == TSR_OK) s = dev_upload(ctx, &n_off, b->gx_off.data(), b->gx_off.size());
    if (s == TSR_OK) s = dev_upload(ctx, &n_gL, b->gx_L.data(), b->gx_L.size());
    if (s == TSR_OK) s = dev_upload(ctx, &n_L, b->L.data(), b->L.size());
    if (s == TSR_OK) s = dev_upload(ctx, &n_cf, cf.data(), cf.size());
    if (s == TSR_OK && mfma_ok) s = dev_upload(ctx, &n_Wm, Wm.data(), Wm.size());
    if (s == TSR_OK && mfma_ok) s = dev_upload(ctx, &n_gm, gm_off.data(), gm_off.size());
    if (s == TSR_OK && mfma_ok) s = dev_upload(ctx, &n_gr, gm_rows.data(), gm_rows.size());
    if (s == TSR_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) {  // host staging vectors die here
        tsr_set_error("bank upload failed");
        s = TSR_ERR_HIP;
    }
    if (s != TSR_OK) {
        dev_free(n_Wx);
        dev_free(n_off);
        dev_free(n_gL);
        dev_free(n_L);
        dev_free(n_cf);
        dev_free(n_Wm);
        dev_free(n_gm);
        dev_free(n_gr);
        return s;
    }
    dev_free(b->d_Wm);
    dev_free(b->d_gm_off);
    dev_free(b->d_gm_rows);
    b->d_Wm = n_Wm;
    b->d_gm_off = n_gm;
    b->d_gm_rows = n_gr;
    b->wm_Dp = Dp;
    b->wm_total4 = 0;
    b->gm_off_h.clear();
    b->gm_rows_h.clear();
    if (mfma_ok) {
        b->gm_off_h = gm_off;
        b->gm_rows_h = gm_rows;
        for (int g = 0; g < b->n_groups; ++g) b->wm_total4 = std::max<int64_t>(b->wm_total4, gm_off[g] + (int64_t)gm_rows[g] * NK4 * 64);
    }
    dev_free(b->d_Wx);
    dev_free(b->d_gx_off);
    dev_free(b->d_gx_L);
    dev_free(b->d_L);
    dev_free(b->d_cf);
    b->d_Wx = n_Wx;
    b->d_gx_off = n_off;
    b->d_gx_L = n_gL;
    b->d_L = n_L;
    b->d_cf = n_cf;
    return TSR_OK;
}

TSR_API int tsr_bank_create(tsr_ctx *ctx, const float *W_cat, const int32_t *L, const double *c,
                            int32_t M, int64_t D, tsr_bank **out)
{
    TSR_REQUIRE(ctx && W_cat && L && c && out, "tsr_bank_create: null pointer");
    *out = nullptr;
    TSR_REQUIRE(M >= 1 && M <= 65535, "tsr_bank_create: M=%d out of range [1,65535]", M);
    TSR_REQUIRE(D >= 1 && D <= (1 << 20), "tsr_bank_create: D=%lld out of range", (long long)D);
    int64_t sumL = 0;
    int Lmax = 0;
    for (int m = 0; m < M; ++m) {
        TSR_REQUIRE(L[m] >= 1 && L[m] <= 65535, "tsr_bank_create: L[%d]=%d out of range [1,65535]", m, L[m]);
        sumL += L[m];
        Lmax = std::max(Lmax, L[m]);
    }
    const int64_t nW = sumL * D;
    for (int64_t i = 0; i < nW; ++i)
        if (!isfinite(W_cat[i])) {
            tsr_set_error("tsr_bank_create: filter value %lld is not finite", (long long)i);
            return TSR_ERR_NONFINITE;
        }
    for (int m = 0; m < M; ++m)
        if (!isfinite(c[m])) {
            tsr_set_error("tsr_bank_create: c[%d] is not finite", m);
            return TSR_ERR_NONFINITE;
        }
    TSR_HIP(hipSetDevice(ctx->device));
    tsr_bank *b = new (std::nothrow) tsr_bank;
    TSR_REQUIRE(b, "tsr_bank_create: out of host memory");
    b->ctx = ctx;
    b->device = ctx->device;
    b->M = M;
    b->D = D;
    b->Lmax = Lmax;
    b->sumL = sumL;
    b->L.assign(L, L + M);
    b->c.assign(c, c + M);
    b->W_cat.assign(W_cat, W_cat + nW);
    int s = bank_build_exact(ctx, b);
    if (s != TSR_OK) {
        tsr_bank_destroy(b);
        return s;
    }
    *out = b;
    return TSR_OK;
}

TSR_API int tsr_bank_destroy(tsr_bank *b)
{
    if (!b) return TSR_OK;
    // never dereference the ctx here: a child may outlive its context (interpreter
    // shutdown destroys objects in arbitrary order); hipFree needs only the device.
    (void)hipSetDevice(b->device);
    (void)hipDeviceSynchronize();
    tsr_fast_bank_free(b);
    dev_free(b->d_Wx);
    dev_free(b->d_Wm);
    dev_free(b->d_gm_off);
    dev_free(b->d_gm_rows);
    dev_free(b->d_gx_off);
    dev_free(b->d_gx_L);
    dev_free(b->d_L);
    dev_free(b->d_cf);
    delete b;
    return TSR_OK;
}

TSR_API int tsr_bank_info(const tsr_bank *b, int32_t *M, int64_t *D, int32_t *Lmax, int64_t *sumL)
{
    TSR_REQUIRE(b, "tsr_bank_info: null bank");
    if (M) *M = b->M;
    if (D) *D = b->D;
    if (Lmax) *Lmax = b->Lmax;
    if (sumL) *sumL = b->sumL;
    return TSR_OK;
}

// ---- corpus ---------------------------------------------------------------------
TSR_API int tsr_corpus_destroy(tsr_corpus *cp);

// Byte rows of a corpus: [P][Dp] + zeroed slack behind the last row (tiles of the scoring kernels may read, never use,
// up to tile_bytes - Dp bytes past it: tsr_staging_read_end).
static int corpus_alloc_bytes(tsr_ctx *ctx, tsr_corpus *cp)
{
    const size_t body = (size_t)cp->P * cp->Dp, slack = (size_t)48 * cp->Dp + 8192;
    if (pool_malloc(cp->device, reinterpret_cast<void **>(&cp->d_feats), body + slack) != hipSuccess ||
        hipMemsetAsync(cp->d_feats + body, 0, slack, ctx->stream) != hipSuccess) {
        tsr_set_error("tsr_corpus_create: cannot allocate %zu bytes of device memory", body + slack);
        return TSR_ERR_HIP;
    }
    cp->feats_alloc = body + slack;
    return TSR_OK;
}

namespace {
__global__ __launch_bounds__(256) void zero_plane_tails_kernel(uint32_t *__restrict__ bits, int64_t plane_dwords, int64_t first, int count)
{
    uint32_t *p = bits + (int64_t)blockIdx.x * plane_dwords + first;
    for (int i = threadIdx.x; i < count; i += 256) p[i] = 0;
}
}  // namespace

int tsr_corpus_alloc_bits(tsr_ctx *ctx, tsr_corpus *cp, bool zero_tails)
{
    tsr_bit_geometry(cp->D, &cp->bit_pw, &cp->bit_nph);
    cp->bit_rows = cp->P + 48;  // (tiles of the bank pass read whole 32-frame tiles: the last one starts at a frame of the corpus)
    const size_t plane = (size_t)cp->bit_rows * cp->bit_pw * 4, total = plane * cp->bit_nph + 256;
    if (pool_malloc(cp->device, reinterpret_cast<void **>(&cp->d_bits), total) != hipSuccess) {
        tsr_set_error("tsr_corpus_part_features: cannot allocate %zu bytes of device memory", total);
        return TSR_ERR_HIP;
    }
    // the read-ahead rows behind the frames of every plane: one launch for all planes (or the caller's setup kernel)
    if (zero_tails) hipLaunchKernelGGL(zero_plane_tails_kernel, dim3((unsigned)cp->bit_nph), dim3(256), 0, ctx->stream, reinterpret_cast<uint32_t *>(cp->d_bits),
                       (int64_t)(plane / 4), (int64_t)cp->P * cp->bit_pw, 48 * cp->bit_pw);
    if (hipGetLastError() != hipSuccess) {
        tsr_set_error("tsr_corpus_part_features: device launch failed");
        return TSR_ERR_HIP;
    }
    cp->bits_alloc = total;
    return TSR_OK;
}

namespace {
// bit rows [P][Bp] -> byte rows [P][Dp] in {0,1}; one lane per 16 output bytes, bytes past D are zero
__global__ __launch_bounds__(256) void corpus_unpack_bits_kernel(const uint8_t *__restrict__ bits, int64_t P, int pw, int64_t rows, int D, int Dp,
                                                                 uint8_t *__restrict__ out)
{
    const int per_row = Dp >> 4;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P * per_row) return;
    const int64_t p = i / per_row;
    const int c = (int)(i - p * per_row);
    const int dw = c >> 1, ph = dw / pw, j = dw - ph * pw;  // dword dw of the frame: plane ph, dword j of its row there
    unsigned b = reinterpret_cast<const uint16_t *>(bits)[(((int64_t)ph * rows + p) * pw + j) * 2 + (c & 1)];
    const int left = D - 16 * c;  // features of this chunk that exist
    if (left < 16) b &= (1u << (left > 0 ? left : 0)) - 1u;
    unsigned w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = (((b >> (4 * q)) & 0xfu) * 0x00204081u) & 0x01010101u;
    *reinterpret_cast<uint4 *>(out + p * Dp + 16 * c) = make_uint4(w[0], w[1], w[2], w[3]);
}
}  // namespace

int tsr_corpus_need_bytes(tsr_ctx *ctx, const tsr_corpus *cpc)
{
    tsr_corpus *cp = const_cast<tsr_corpus *>(cpc);  // a derived form, cached (corpora are immutable once created)
    if (cp->d_feats || cp->P == 0) return TSR_OK;
    TSR_REQUIRE(cp->d_bits, "corpus holds no features");
    TSR_HIP(hipSetDevice(ctx->device));
    TSR_TRY(corpus_alloc_bytes(ctx, cp));
    const int64_t n = cp->P * (cp->Dp >> 4);
    hipLaunchKernelGGL(corpus_unpack_bits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, cp->d_bits, cp->P,
                       cp->bit_pw, cp->bit_rows, (int)cp->D, cp->Dp, cp->d_feats);
    TSR_HIP(hipGetLastError());
    return TSR_OK;
}

namespace {
// frame p -> [first frame, end frame) of the utterance that holds it (the last n with offs[n] <= p)
__global__ __launch_bounds__(256) void utt_tables_kernel(const int64_t *__restrict__ offs, int64_t N, int64_t P, int32_t *__restrict__ beg,
                                                         int32_t *__restrict__ end)
{
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) {
        if (p < P + 256) end[p] = 0;  // the 256 ints behind utt_end that tiles read past P
        return;
    }
    int64_t lo = 0, hi = N;  // invariant: offs[lo] <= p < offs[hi]  (empty utterances share an offset: the last one wins, its end > p)
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (offs[mid] <= p)
            lo = mid;
        else
            hi = mid;
    }
    beg[p] = (int32_t)offs[lo];
    end[p] = (int32_t)offs[lo + 1];
}
}  // namespace

// Allocate a corpus (resident feature rows zero-initialised only in the slack, utterance tables
// made on the device, result arrays allocated) without any feature data.
int tsr_corpus_create_empty(tsr_ctx *ctx, const int64_t *frame_offsets, int64_t N, int64_t D,
                            tsr_corpus **out, bool alloc_bytes, const int64_t *d_off_ready, bool tables)
{
    TSR_REQUIRE(ctx && frame_offsets && out, "tsr_corpus_create: null pointer");
    *out = nullptr;
    TSR_REQUIRE(N >= 0 && D >= 1, "tsr_corpus_create: bad N=%lld or D=%lld", (long long)N, (long long)D);
    TSR_REQUIRE(frame_offsets[0] == 0, "tsr_corpus_create: frame_offsets[0] must be 0");
    for (int64_t n = 0; n < N; ++n)
        TSR_REQUIRE(frame_offsets[n + 1] >= frame_offsets[n], "tsr_corpus_create: frame_offsets not monotone at %lld", (long long)n);
    const int64_t P = frame_offsets[N];
    TSR_REQUIRE(P < (int64_t)1 << 31, "tsr_corpus_create: %lld frames exceed 2^31-1 per corpus", (long long)P);
    TSR_HIP(hipSetDevice(ctx->device));
    tsr_corpus *cp = new (std::nothrow) tsr_corpus;
    TSR_REQUIRE(cp, "tsr_corpus_create: out of host memory");
    cp->ctx = ctx;
    cp->device = ctx->device;
    cp->N = N;
    cp->D = D;
    cp->P = P;
    cp->offsets.assign(frame_offsets, frame_offsets + N + 1);
    // resident form: 0/1 bytes, rows of Dp = round_up(D,16) bytes, zeroed slack behind the
    // last row (tiles of the scoring kernels may read, never use, up to ~48 rows past it)
    cp->Dp = (int)((D + 15) / 16 * 16);
    int s = alloc_bytes ? corpus_alloc_bytes(ctx, cp) : TSR_OK;
    // per-frame utterance tables (first frame / end frame of the utterance holding p; 256 zero ints behind utt_end: tiles
    // read past P), made on the device from the N + 1 offsets; scores and ids are written before they are read
    auto grab = [&](void **q, size_t bytes) { return pool_malloc(cp->device, q, bytes) == hipSuccess ? TSR_OK : TSR_ERR_HIP; };
    if (s == TSR_OK) s = grab(reinterpret_cast<void **>(&cp->d_utt_beg), (size_t)std::max<int64_t>(P, 1) * 4);
    if (s == TSR_OK) s = grab(reinterpret_cast<void **>(&cp->d_utt_end), (size_t)(P + 256) * 4);
    if (s == TSR_OK) s = grab(reinterpret_cast<void **>(&cp->d_scores), (size_t)std::max<int64_t>(P, 1) * 4);
    if (s == TSR_OK) s = grab(reinterpret_cast<void **>(&cp->d_ids), (size_t)std::max<int64_t>(P, 1) * 4);
    // (d_off_ready: the caller has the N + 1 offsets on the device already, enqueued on the stream, and synchronises itself)
    const void *d_off = d_off_ready;
    if (s == TSR_OK && !d_off_ready && tables) {
        void *q = nullptr;
        s = tsr_scratch(ctx, SLOT_CNT, (size_t)(N + 1) * 8 + 64, &q);
        d_off = q;
        if (s == TSR_OK && hipMemcpyAsync(q, frame_offsets, (size_t)(N + 1) * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) s = TSR_ERR_HIP;
    }
    if (s == TSR_OK && tables) {  // (the grid covers the zeroed slack behind utt_end as well)
        hipLaunchKernelGGL(utt_tables_kernel, dim3((unsigned)((P + 256 + 255) / 256)), dim3(256), 0, ctx->stream, (const int64_t *)d_off, N, P,
                           cp->d_utt_beg, cp->d_utt_end);
        if (hipGetLastError() != hipSuccess) s = TSR_ERR_HIP;
    }
    if (s == TSR_ERR_HIP) tsr_set_error("tsr_corpus_create: cannot allocate or fill the corpus tables");
    if (s == TSR_OK && !d_off_ready && hipStreamSynchronize(ctx->stream) != hipSuccess) {
        tsr_set_error("tsr_corpus_create: upload failed");
        s = TSR_ERR_HIP;
    }
    if (s != TSR_OK) {
        tsr_corpus_destroy(cp);
        return s;
    }
    *out = cp;
    return TSR_OK;
}

TSR_API int tsr_corpus_create(tsr_ctx *ctx, const uint8_t *feats, const int64_t *frame_offsets,
                              int64_t N, int64_t D, tsr_corpus **out)
{
    TSR_REQUIRE(out, "tsr_corpus_create: null pointer");
    tsr_corpus *cp = nullptr;
    TSR_TRY(tsr_corpus_create_empty(ctx, frame_offsets, N, D, &cp));
    const int64_t P = cp->P;
    int s = TSR_OK;
    if (P > 0) {
        if (!feats) {
            tsr_set_error("tsr_corpus_create: null feats");
            s = TSR_ERR_INVALID;
        }
        void *stage = nullptr;
        if (s == TSR_OK) s = tsr_scratch(ctx, SLOT_IN, (size_t)(P * D), &stage);
        if (s == TSR_OK && hipMemcpyAsync(stage, feats, (size_t)(P * D), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) s = TSR_ERR_HIP;
        if (s == TSR_OK) s = tsr_launch_corpus_pack(ctx, (const uint8_t *)stage, P, (int)D, cp->Dp, cp->d_feats);
        if (s == TSR_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) s = TSR_ERR_HIP;
        if (s == TSR_ERR_HIP) tsr_set_error("tsr_corpus_create: upload failed");
    }
    if (s != TSR_OK) {
        tsr_corpus_destroy(cp);
        return s;
    }
    *out = cp;
    return TSR_OK;
}

TSR_API int tsr_corpus_destroy(tsr_corpus *cp)
{
    if (!cp) return TSR_OK;
    // never dereference the ctx here: a child may outlive its context (interpreter
    // shutdown destroys objects in arbitrary order); hipFree needs only the device.
    (void)hipSetDevice(cp->device);
    (void)hipDeviceSynchronize();
    pool_free(cp->d_feats);
    pool_free(cp->d_bits);
    pool_free(cp->d_utt_beg);
    pool_free(cp->d_utt_end);
    pool_free(cp->d_scores);
    pool_free(cp->d_ids);
    dev_free(cp->d_plan);
    dev_free(cp->d_aff);
    dev_free(cp->d_feats_bf16);
    dev_free(cp->d_planes);
    tsr_mx_corpus_free(cp);
    delete cp;
    return TSR_OK;
}

TSR_API int tsr_corpus_frames(const tsr_corpus *cp, int64_t *total_frames)
{
    TSR_REQUIRE(cp && total_frames, "tsr_corpus_frames: null pointer");
    *total_frames = cp->P;
    return TSR_OK;
}

// One-off probe per context: the bit-identity of exact_mfma_kernel with the reference's sequential f32
// sums rests on how this chip's v_mfma_f32_16x16x4_f32 rounds (an ascending fmaf chain, one rounding per step) and on a
// compiler that keeps the hazard distance in front of its operand selects.  Before the first TSR_EXACT pass goes to the
// matrix cores, a small fixed bank and corpus (two template groups, lengths 1 .. 23, D off the 16-byte grid, ragged
// utterances) are scored by both exact kernels and the raw sums compared bit for bit; on any difference the context
// keeps to the vector kernel from then on.
static int exact_mfma_probe(tsr_ctx *ctx)
{
    if (ctx->exact_mfma_state != 0) return TSR_OK;
    ctx->exact_mfma_state = -1;  // (until proven otherwise; also stops recursion)
    const int M = 11, D = 45, N = 4;
    const int64_t Ts[N] = {70, 33, 41, 9};
    std::vector<int32_t> L(M);
    std::vector<double> c(M, 0.0);
    std::vector<int64_t> offs(N + 1, 0);
    for (int n = 0; n < N; ++n) offs[n + 1] = offs[n] + Ts[n];
    int64_t sumL = 0;
    for (int m = 0; m < M; ++m) sumL += (L[m] = 1 + (m * 7) % 23);
    uint32_t x = 0x9e3779b9u;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; };
    std::vector<float> W((size_t)sumL * D);
    for (float &w : W) {  // magnitudes over ten binades, both signs: sums that round at every step
        const uint32_t r = rnd();
        w = ldexpf(1.0f + (float)(r & 0x7fffff) / 8388608.0f, (int)((r >> 23) & 7) - 6) * ((r >> 31) ? -1.0f : 1.0f);
    }
    std::vector<uint8_t> F((size_t)offs[N] * D);
    for (uint8_t &b : F) b = (rnd() & 3) == 0;
    tsr_bank *bank = nullptr;
    tsr_corpus *cp = nullptr;
    int s = tsr_bank_create(ctx, W.data(), L.data(), c.data(), M, D, &bank);
    if (s == TSR_OK) s = tsr_corpus_create(ctx, F.data(), offs.data(), N, D, &cp);
    void *raw = nullptr;
    const size_t n = (size_t)M * (size_t)offs[N];
    std::vector<float> a(n), b(n);
    if (s == TSR_OK) s = tsr_scratch(ctx, SLOT_RAW, 2 * n * sizeof(float), &raw);
    bool same = false;
    if (s == TSR_OK) {
        float *r0 = (float *)raw, *r1 = r0 + n;
        s = tsr_launch_exact_bank_mfma(ctx, bank, cp, r0);
        if (s == TSR_OK) s = tsr_launch_exact_bank(ctx, bank, cp->d_feats, cp->P, cp->Dp, r1);
        if (s == TSR_OK && (hipMemcpyAsync(a.data(), r0, n * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                            hipMemcpyAsync(b.data(), r1, n * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                            hipStreamSynchronize(ctx->stream) != hipSuccess))
            s = TSR_ERR_HIP;
        same = s == TSR_OK && memcmp(a.data(), b.data(), n * 4) == 0;
    }
    if (cp) tsr_corpus_destroy(cp);
    if (bank) tsr_bank_destroy(bank);
    if (same) {
        ctx->exact_mfma_state = 1;
        ctx->exact_mfma_could_not_run = false;
    } else if (s == TSR_OK) {  // both kernels ran and the sums differ: the only case that is latched
        fprintf(stderr, "libtsr_hip: TSR_EXACT stays on the vector kernel on device %d: the matrix-core probe did not reproduce the "
                        "sequential f32 sums bit for bit\n", ctx->device);
    } else {  // a transient failure (allocation, launch): this pass takes the vector kernel, the next exact pass probes again
        if (!ctx->exact_mfma_could_not_run)
            fprintf(stderr, "libtsr_hip: the matrix-core probe of TSR_EXACT could not run on device %d (%s); this pass uses the vector "
                            "kernel, the next one probes again\n", ctx->device, tsr_last_error());
        ctx->exact_mfma_state = 0;
        ctx->exact_mfma_could_not_run = true;
        tsr_set_error("%s", "");  // the probe's failure is not the caller's: TSR_OK is returned with a clean error string
    }
    return TSR_OK;  // (never an error of the caller's pass: the vector kernel is always there)
}

static int corpus_score_impl(tsr_ctx *ctx, const tsr_bank *bank, tsr_corpus *cp, int flags,
                             float *d_all)
{
    TSR_REQUIRE(ctx && bank && cp, "tsr_corpus_score: null pointer");
    TSR_REQUIRE(bank->D == cp->D, "tsr_corpus_score: bank D=%lld but corpus D=%lld", (long long)bank->D, (long long)cp->D);
    TSR_REQUIRE((flags & ~(TSR_FAST | TSR_W24 | TSR_BF16 | TSR_MX)) == 0, "tsr_corpus_score: unknown flags 0x%x", flags);
    TSR_REQUIRE(!(flags & (TSR_W24 | TSR_BF16 | TSR_MX)) || (flags & TSR_FAST), "tsr_corpus_score: TSR_W24 / TSR_BF16 / TSR_MX modify TSR_FAST");
    TSR_REQUIRE(!(flags & TSR_MX) || !(flags & (TSR_W24 | TSR_BF16)), "tsr_corpus_score: TSR_MX stands alone beside TSR_FAST");
    if (cp->P == 0) return TSR_OK;
    TSR_HIP(hipSetDevice(ctx->device));
    if (flags & TSR_MX) {
        // 13-bit weights as three fp6 limbs on the block-scaled matrix instruction (detect_mx.hip): banks of up to eight
        // templates on rows of up to 512 features (max mode and all-scores passes).  Anything else -- and short filters,
        // whose scores of order one want more than 13 bits -- is served as plain TSR_FAST.
        flags &= ~TSR_MX;
        int32_t Lmin = bank->Lmax;
        for (int32_t l : bank->L) Lmin = std::min(Lmin, l);
        if ((int64_t)Lmin * bank->D > TSR_FAST_W24_BELOW_TAPS) {
            const int sm = tsr_launch_mx_bank(ctx, bank, cp, d_all ? nullptr : cp->d_scores, d_all ? nullptr : cp->d_ids, d_all);
            if (sm == TSR_OK) ctx->last_bank_kernel = TSR_KERNEL_MX;
            if (sm != TSR_ERR_UNSUPPORTED) return sm;
        }
    }
    // Short filters (few thousand taps) produce scores of order one, against which the 16-bit
    // quantisation noise is no longer small; their launch is latency-bound anyway, so TSR_FAST
    // takes the third limb there at no measurable cost (include/tsr.h, "at least 16-bit weights").
    if ((flags & TSR_FAST) && !(flags & TSR_BF16)) {
        int32_t Lmin = bank->Lmax;
        for (int32_t l : bank->L) Lmin = std::min(Lmin, l);
        if ((int64_t)Lmin * bank->D <= TSR_FAST_W24_BELOW_TAPS) flags |= TSR_W24;  // (the shortest filter decides)
        // Very long filters: 16-bit weights times more than 66 052 set taps no longer fit the two-limb
        // path's int32 window sums; rather than shave bits off the weights (which would break the
        // "at least 16-bit" bound of include/tsr.h) the bank takes three limbs and the int64 ring.
        if ((int64_t)bank->Lmax * bank->D > TSR_FAST_W24_ABOVE_TAPS) flags |= TSR_W24;
    }
    if (flags == TSR_FAST && !bank->tap2_unsupported) {
        // Two limbs on narrow feature rows (an odd number <= 23 of 16-byte slots: the edge features of BASELINE
        // configs 1, 2 and 4): the tap-pair kernel (detect_mfma_tap2.inc: two taps folded into K, half the skew
        // emission per matrix instruction) -- config 2: 7 % faster than the interleaved kernel, config 4: 10 %.
        // Wider rows, and filters too long for its LDS ring, stay on the interleaved kernel.  The context option
        // TSR_OPT_FAST_KERNEL forces one or the other (A/B measurements inside one process).
        bool want = ctx->opt_fast_kernel == TSR_FAST_KERNEL_TAP2;
        if (ctx->opt_fast_kernel == TSR_OPT_AUTO) {
            const int nkb2 = cp->Dp / 16;
            want = (nkb2 & 1) && nkb2 <= 23;
        }
        if (want) {
            tsr_bank *mb = const_cast<tsr_bank *>(bank);
            int sb = bank->fast[4] ? TSR_OK : tsr_fast_bank_build(ctx, mb, 4);
            if (sb == TSR_OK) sb = tsr_launch_tap2_bank(ctx, bank, cp, d_all ? nullptr : cp->d_scores, d_all ? nullptr : cp->d_ids, d_all);
            if (sb == TSR_OK) ctx->last_bank_kernel = TSR_KERNEL_TAP2;
            if (sb != TSR_ERR_UNSUPPORTED) return sb;
            if (!bank->fast[4]) mb->tap2_unsupported = true;   // geometry never fits: do not try again
        }
    }
    if (flags & TSR_FAST) {
        const int kind = ((flags & TSR_BF16) ? 2 : 0) + ((flags & TSR_W24) ? 1 : 0);
        if (!bank->fast[kind]) {
            int sb = tsr_fast_bank_build(ctx, const_cast<tsr_bank *>(bank), kind);
            if (sb != TSR_OK && sb != TSR_ERR_UNSUPPORTED) return sb;
        }
        int s = tsr_launch_fast_bank(ctx, bank, cp, flags, d_all ? nullptr : cp->d_scores,
                                     d_all ? nullptr : cp->d_ids, d_all);
        if (s == TSR_OK) ctx->last_bank_kernel = TSR_KERNEL_INTERLEAVED;
        if (s != TSR_ERR_UNSUPPORTED) return s;
        // geometry outside the MFMA kernel's envelope: the exact GPU kernel serves it
    }
    if (ctx->opt_exact_kernel != TSR_EXACT_KERNEL_VALU) TSR_TRY(exact_mfma_probe(ctx));  // (first exact pass of this context only)
    void *raw = nullptr;
    TSR_TRY(tsr_scratch(ctx, SLOT_RAW, (size_t)bank->M * cp->P * sizeof(float), &raw));
    // the reference's sums bit for bit: on the matrix cores where the bank has that form (the context option
    // TSR_OPT_EXACT_KERNEL = TSR_EXACT_KERNEL_VALU keeps the vector kernel, for A/B measurements), otherwise one lane
    // per window
    int se = TSR_ERR_UNSUPPORTED;
    if (ctx->opt_exact_kernel != TSR_EXACT_KERNEL_VALU && ctx->exact_mfma_state > 0)
        se = tsr_launch_exact_bank_mfma(ctx, bank, cp, (float *)raw);
    ctx->last_bank_kernel = se == TSR_ERR_UNSUPPORTED ? TSR_KERNEL_EXACT_VALU : TSR_KERNEL_EXACT_MFMA;
    if (se == TSR_ERR_UNSUPPORTED) {
        TSR_TRY(tsr_corpus_need_bytes(ctx, cp));
        se = tsr_launch_exact_bank(ctx, bank, cp->d_feats, cp->P, cp->Dp, (float *)raw);
    }
    TSR_TRY(se);
    return tsr_launch_bank_epilogue(ctx, bank, (const float *)raw, cp->P, cp->d_utt_end, flags,
                                    d_all ? nullptr : cp->d_scores, d_all ? nullptr : cp->d_ids, d_all);
}

TSR_API int tsr_corpus_score(tsr_ctx *ctx, const tsr_bank *bank, tsr_corpus *cp, int flags)
{
    if (cp) cp->scored = false;  // (a failed pass may have written part of the arrays)
    const int s = corpus_score_impl(ctx, bank, cp, flags, nullptr);
    if (s == TSR_OK && cp) cp->scored = true;
    return s;
}

TSR_API int tsr_corpus_score_all(tsr_ctx *ctx, const tsr_bank *bank, tsr_corpus *cp, int flags, float *per_template)
{
    TSR_REQUIRE(ctx && bank && cp && (per_template || cp->P == 0), "tsr_corpus_score_all: null pointer");
    if (cp->P == 0) return TSR_OK;
    TSR_HIP(hipSetDevice(ctx->device));
    void *d_all = nullptr;
    const size_t bytes = (size_t)bank->M * cp->P * sizeof(float);
    TSR_TRY(tsr_scratch(ctx, SLOT_OUT, bytes, &d_all));
    TSR_TRY(corpus_score_impl(ctx, bank, cp, flags, (float *)d_all));
    return tsr_download(ctx, per_template, d_all, bytes);
}

TSR_API int tsr_corpus_download(tsr_ctx *ctx, const tsr_corpus *cp, float *scores, int32_t *ids)
{
    TSR_REQUIRE(ctx && cp, "tsr_corpus_download: null pointer");
    TSR_REQUIRE(cp->scored || cp->P == 0, "tsr_corpus_download: this corpus has not been scored (tsr_corpus_score) -- its result arrays hold nothing yet");
    TSR_HIP(hipSetDevice(ctx->device));
    if (scores && cp->P) TSR_TRY(tsr_download(ctx, scores, cp->d_scores, (size_t)cp->P * 4));
    if (ids && cp->P) TSR_TRY(tsr_download(ctx, ids, cp->d_ids, (size_t)cp->P * 4));
    TSR_HIP(hipStreamSynchronize(ctx->stream));
    return TSR_OK;
}

TSR_API int tsr_score_bank_u8(tsr_ctx *ctx, const tsr_bank *bank, const uint8_t *feats,
                              const int64_t *frame_offsets, int64_t N, int flags, float *scores,
                              int32_t *ids)
{
    TSR_REQUIRE(ctx && bank, "tsr_score_bank_u8: null pointer");
    tsr_corpus *cp = nullptr;
    TSR_TRY(tsr_corpus_create(ctx, feats, frame_offsets, N, bank->D, &cp));
    int s = tsr_corpus_score(ctx, bank, cp, flags);
    if (s == TSR_OK) s = tsr_corpus_download(ctx, cp, scores, ids);
    tsr_corpus_destroy(cp);
    return s;
}

TSR_API int tsr_score_bank_all_u8(tsr_ctx *ctx, const tsr_bank *bank, const uint8_t *feats,
                                  const int64_t *frame_offsets, int64_t N, int flags,
                                  float *per_template)
{
    TSR_REQUIRE(ctx && bank && per_template, "tsr_score_bank_all_u8: null pointer");
    tsr_corpus *cp = nullptr;
    TSR_TRY(tsr_corpus_create(ctx, feats, frame_offsets, N, bank->D, &cp));
    int s = TSR_OK;
    void *d_all = nullptr;
    const size_t bytes = (size_t)bank->M * cp->P * sizeof(float);
    if (cp->P) {
        s = tsr_scratch(ctx, SLOT_OUT, bytes, &d_all);
        if (s == TSR_OK) s = corpus_score_impl(ctx, bank, cp, flags, (float *)d_all);
        if (s == TSR_OK && hipMemcpyAsync(per_template, d_all, bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) s = TSR_ERR_HIP;
        if (s == TSR_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) s = TSR_ERR_HIP;
        if (s == TSR_ERR_HIP) tsr_set_error("tsr_score_bank_all_u8: device copy failed");
    }
    tsr_corpus_destroy(cp);
    return s;
}

// ---- phone-window classification -----------------------------------
// get_train_data.py:718-773 over a resident corpus: windows are [start, end) frame ranges of the
// stream, each inside one utterance.
TSR_API int tsr_corpus_classify_windows(tsr_ctx *ctx, const tsr_bank *bank, tsr_corpus *cp, const int64_t *starts,
                                        const int64_t *ends, int64_t W, int flags, float *score, int64_t *loc,
                                        int32_t *template_id)
{
    TSR_REQUIRE(ctx && bank && cp && (W == 0 || (starts && ends && score && loc && template_id)),
                "tsr_corpus_classify_windows: null pointer");
    TSR_REQUIRE(W >= 0, "tsr_corpus_classify_windows: negative window count");
    if (W == 0) return TSR_OK;
    for (int64_t w = 0; w < W; ++w) {
        TSR_REQUIRE(starts[w] >= 0 && starts[w] < ends[w] && ends[w] <= cp->P,
                    "tsr_corpus_classify_windows: window %lld = [%lld, %lld) is empty or outside the %lld frames (np.argmax of an empty slice raises in the reference)",
                    (long long)w, (long long)starts[w], (long long)ends[w], (long long)cp->P);
        const auto it = std::upper_bound(cp->offsets.begin(), cp->offsets.end(), starts[w]);  // first offset > start
        TSR_REQUIRE(it != cp->offsets.end() && ends[w] <= *it,
                    "tsr_corpus_classify_windows: window %lld = [%lld, %lld) crosses an utterance boundary",
                    (long long)w, (long long)starts[w], (long long)ends[w]);
    }
    TSR_HIP(hipSetDevice(ctx->device));
    void *d_all = nullptr, *d_win = nullptr, *d_res = nullptr;
    TSR_TRY(tsr_scratch(ctx, SLOT_OUT, (size_t)bank->M * cp->P * sizeof(float), &d_all));
    TSR_TRY(tsr_scratch(ctx, SLOT_AUX, (size_t)W * 16, &d_win));
    TSR_TRY(tsr_scratch(ctx, SLOT_AUX2, (size_t)W * 16, &d_res));
    int64_t *d_starts = (int64_t *)d_win, *d_ends = d_starts + W;
    int64_t *d_loc = (int64_t *)d_res;
    float *d_score = (float *)(d_loc + W);
    int32_t *d_tid = (int32_t *)(d_score + W);
    TSR_HIP(hipMemcpyAsync(d_starts, starts, (size_t)W * 8, hipMemcpyHostToDevice, ctx->stream));
    TSR_HIP(hipMemcpyAsync(d_ends, ends, (size_t)W * 8, hipMemcpyHostToDevice, ctx->stream));
    TSR_TRY(corpus_score_impl(ctx, bank, cp, flags, (float *)d_all));
    TSR_TRY(tsr_launch_window_argmax(ctx, (const float *)d_all, cp->P, (int)bank->M, d_starts, d_ends, W, d_score, d_loc, d_tid));
    TSR_HIP(hipMemcpyAsync(score, d_score, (size_t)W * 4, hipMemcpyDeviceToHost, ctx->stream));
    TSR_HIP(hipMemcpyAsync(loc, d_loc, (size_t)W * 8, hipMemcpyDeviceToHost, ctx->stream));
    TSR_HIP(hipMemcpyAsync(template_id, d_tid, (size_t)W * 4, hipMemcpyDeviceToHost, ctx->stream));
    TSR_HIP(hipStreamSynchronize(ctx->stream));
    return TSR_OK;
}

// get_train_data.py:838-925: W background-padded f32 windows, rows concatenated in `E_cat`
// (window w = rows [row_offsets[w], row_offsets[w+1])), arg-max over its first n_valid[w] positions.
TSR_API int tsr_classify_float_windows(tsr_ctx *ctx, const tsr_bank *bank, const float *E_cat,
                                       const int64_t *row_offsets, const int32_t *n_valid, int64_t W,
                                       float *score, int32_t *loc, int32_t *template_id)
{
    TSR_REQUIRE(ctx && bank && (W == 0 || (E_cat && row_offsets && n_valid && score && loc && template_id)),
                "tsr_classify_float_windows: null pointer");
    TSR_REQUIRE(W >= 0, "tsr_classify_float_windows: negative window count");
    if (W == 0) return TSR_OK;
    TSR_REQUIRE(row_offsets[0] == 0, "tsr_classify_float_windows: row_offsets[0] must be 0");
    std::vector<int64_t> starts((size_t)W), ends((size_t)W);
    int64_t NW = 0;
    for (int64_t w = 0; w < W; ++w) {
        const int64_t T = row_offsets[w + 1] - row_offsets[w];
        TSR_REQUIRE(T >= 1 && n_valid[w] >= 1 && n_valid[w] <= T,
                    "tsr_classify_float_windows: window %lld has %lld rows and %d positions to search (np.argmax of an empty slice raises in the reference)",
                    (long long)w, (long long)T, n_valid[w]);
        for (int32_t m = 0; m < bank->M; ++m)
            TSR_REQUIRE(T >= bank->L[m], "tsr_classify_float_windows: window %lld has %lld rows, template %d needs %d (the reference reads out of bounds here)",
                        (long long)w, (long long)T, m, bank->L[m]);
        starts[(size_t)w] = NW;
        NW += n_valid[w];
        ends[(size_t)w] = NW;
    }
    const int64_t R = row_offsets[W];
    TSR_REQUIRE(R < ((int64_t)1 << 31) && NW < ((int64_t)1 << 31), "tsr_classify_float_windows: too many rows");
    std::vector<int32_t> item_row((size_t)NW), item_left((size_t)NW);
    for (int64_t w = 0; w < W; ++w)
        for (int32_t t = 0; t < n_valid[w]; ++t) {
            item_row[(size_t)(starts[(size_t)w] + t)] = (int32_t)(row_offsets[w] + t);
            item_left[(size_t)(starts[(size_t)w] + t)] = (int32_t)(row_offsets[w + 1] - row_offsets[w] - t);
        }
    TSR_HIP(hipSetDevice(ctx->device));
    void *dX = nullptr, *d_all = nullptr, *d_win = nullptr, *d_res = nullptr, *d_items = nullptr;
    TSR_TRY(tsr_scratch(ctx, SLOT_IN, (size_t)R * bank->D * 4, &dX));
    TSR_TRY(tsr_scratch(ctx, SLOT_OUT, (size_t)bank->M * NW * 4, &d_all));
    TSR_TRY(tsr_scratch(ctx, SLOT_AUX, (size_t)W * 16, &d_win));
    TSR_TRY(tsr_scratch(ctx, SLOT_AUX2, (size_t)W * 16, &d_res));
    TSR_TRY(tsr_scratch(ctx, SLOT_W, (size_t)NW * 8, &d_items));
    int64_t *d_starts = (int64_t *)d_win, *d_ends = d_starts + W;
    int64_t *d_loc = (int64_t *)d_res;
    float *d_score = (float *)(d_loc + W);
    int32_t *d_tid = (int32_t *)(d_score + W);
    int32_t *d_row = (int32_t *)d_items, *d_left = d_row + NW;
    TSR_HIP(hipMemcpyAsync(dX, E_cat, (size_t)R * bank->D * 4, hipMemcpyHostToDevice, ctx->stream));
    TSR_HIP(hipMemcpyAsync(d_starts, starts.data(), (size_t)W * 8, hipMemcpyHostToDevice, ctx->stream));
    TSR_HIP(hipMemcpyAsync(d_ends, ends.data(), (size_t)W * 8, hipMemcpyHostToDevice, ctx->stream));
    TSR_HIP(hipMemcpyAsync(d_row, item_row.data(), (size_t)NW * 4, hipMemcpyHostToDevice, ctx->stream));
    TSR_HIP(hipMemcpyAsync(d_left, item_left.data(), (size_t)NW * 4, hipMemcpyHostToDevice, ctx->stream));
    TSR_TRY(tsr_launch_float_windows(ctx, bank, (const float *)dX, d_row, d_left, NW, (float *)d_all));
    TSR_TRY(tsr_launch_window_argmax(ctx, (const float *)d_all, NW, (int)bank->M, d_starts, d_ends, W, d_score, d_loc, d_tid));
    std::vector<int64_t> loc64((size_t)W);
    TSR_HIP(hipMemcpyAsync(score, d_score, (size_t)W * 4, hipMemcpyDeviceToHost, ctx->stream));
    TSR_HIP(hipMemcpyAsync(loc64.data(), d_loc, (size_t)W * 8, hipMemcpyDeviceToHost, ctx->stream));
    TSR_HIP(hipMemcpyAsync(template_id, d_tid, (size_t)W * 4, hipMemcpyDeviceToHost, ctx->stream));
    TSR_HIP(hipStreamSynchronize(ctx->stream));
    for (int64_t w = 0; w < W; ++w) loc[w] = (int32_t)(loc64[(size_t)w] - starts[(size_t)w]);  // position inside the window
    return TSR_OK;
}

// get_train_data.py:781-925 over a resident binary corpus, the windows never materialised: window w
// starts at stream frame first[w] (may lie before its utterance: front padding), its rows are real
// corpus frames inside [real_lo[w], real_hi[w]) and the background vector everywhere else, it has
// rows[w] rows in total and its first n_valid[w] positions are searched.
TSR_API int tsr_corpus_classify_isolated(tsr_ctx *ctx, const tsr_bank *bank, const tsr_corpus *cp, const float *bgd,
                                         const int64_t *first, const int64_t *real_lo, const int64_t *real_hi,
                                         const int32_t *rows, const int32_t *n_valid, int64_t W, float *score,
                                         int32_t *loc, int32_t *template_id)
{
    TSR_REQUIRE(ctx && bank && cp && bgd && (W == 0 || (first && real_lo && real_hi && rows && n_valid && score && loc && template_id)),
                "tsr_corpus_classify_isolated: null pointer");
    TSR_REQUIRE(bank->D == cp->D, "tsr_corpus_classify_isolated: bank D=%lld but corpus D=%lld", (long long)bank->D, (long long)cp->D);
    TSR_REQUIRE(W >= 0, "tsr_corpus_classify_isolated: negative window count");
    if (W == 0) return TSR_OK;
    std::vector<int64_t> starts((size_t)W), ends((size_t)W);
    int64_t NW = 0;
    for (int64_t w = 0; w < W; ++w) {
        TSR_REQUIRE(rows[w] >= 1 && n_valid[w] >= 1 && n_valid[w] <= rows[w],
                    "tsr_corpus_classify_isolated: window %lld has %d rows and %d positions to search (np.argmax of an empty slice raises in the reference)",
                    (long long)w, rows[w], n_valid[w]);
        TSR_REQUIRE(real_lo[w] >= 0 && real_hi[w] <= cp->P && real_lo[w] <= real_hi[w],
                    "tsr_corpus_classify_isolated: window %lld: real frames [%lld, %lld) outside the corpus", (long long)w, (long long)real_lo[w], (long long)real_hi[w]);
        for (int32_t m = 0; m < bank->M; ++m)
            TSR_REQUIRE(rows[w] >= bank->L[m], "tsr_corpus_classify_isolated: window %lld has %d rows, template %d needs %d (the reference reads out of bounds here)",
                        (long long)w, rows[w], m, bank->L[m]);
        starts[(size_t)w] = NW;
        NW += n_valid[w];
        ends[(size_t)w] = NW;
    }
    TSR_REQUIRE(NW < ((int64_t)1 << 30), "tsr_corpus_classify_isolated: too many positions");
    for (int64_t d = 0; d < bank->D; ++d) TSR_REQUIRE(std::isfinite(bgd[d]), "tsr_corpus_classify_isolated: background value %lld is not finite", (long long)d);
    TSR_HIP(hipSetDevice(ctx->device));
    // the scores of every searched position: on the matrix cores where the bank has that form, this context's probe of the
    // f32 chains matched (run here, before any scratch buffer of this call exists: it uses them itself) and the option
    // does not ask for the vector kernel; else one lane per position
    bool on_mfma = false;
    int iso_rmax = 1;  // physical rows a pair of positions can span: the longest template group's rows
    for (int32_t r : bank->gm_rows_h) iso_rmax = std::max(iso_rmax, (int)r);
    if (bank->d_Wm && cp->Dp == bank->wm_Dp && ctx->opt_exact_kernel != TSR_EXACT_KERNEL_VALU &&
        (iso_rmax + 1) * (iso_rmax + 1) <= TSR_ISO_MAX_BUCKETS) {
        TSR_TRY(exact_mfma_probe(ctx));
        on_mfma = ctx->exact_mfma_state > 0;
    }
    TSR_TRY(tsr_corpus_need_bytes(ctx, cp));
    // Everything the kernels read goes up in ONE copy (background vector | window starts | window ends | tables of the
    // kernel) and the three result arrays come back in one: a pageable copy costs its fixed 10-20 us whatever its size.
    void *d_all = nullptr, *d_res = nullptr, *d_up = nullptr;
    TSR_TRY(tsr_scratch(ctx, SLOT_OUT, (size_t)bank->M * NW * 4, &d_all));
    TSR_TRY(tsr_scratch(ctx, SLOT_AUX2, (size_t)W * 16, &d_res));
    int64_t *d_loc = (int64_t *)d_res;
    float *d_score = (float *)(d_loc + W);
    int32_t *d_tid = (int32_t *)(d_score + W);
    const size_t off_bgd = 0, off_starts = ((size_t)bank->D * 4 + 15) / 16 * 16, off_ends = off_starts + (size_t)W * 8,
                 off_tab = off_ends + (size_t)W * 8;  // (all multiples of 8; the tables start 16-byte aligned: W * 16 behind off_starts)
    std::vector<uint8_t> up;   // (kept alive until the stream is synchronised below)
    auto put = [&](size_t off, const void *src, size_t bytes) { memcpy(up.data() + off, src, bytes); };
    if (on_mfma) {
        // pairs of consecutive positions, bucketed by padding pattern (physical rows [k, k + r) of the pair real) ON THE DEVICE:
        // the windows' geometry goes up (first | real_lo | real_hi | starts | pair prefix | rows | n_valid), four small
        // kernels count, scan, scatter and fill the tiles (detect_exact_mfma.hip); the matrix-core kernel's grid covers the
        // upper bound of the tile count and reads the count itself
        const int RMAX = iso_rmax;
        const int NB = (RMAX + 1) * (RMAX + 1);
        const size_t off_geo = off_tab;   // (16-byte aligned)
        const size_t geo_bytes = (size_t)(5 * W + 1) * 8 + (size_t)2 * W * 4;
        up.resize(off_geo + geo_bytes + 16);
        int64_t *gq = reinterpret_cast<int64_t *>(up.data() + off_geo);
        memcpy(gq, first, (size_t)W * 8);
        memcpy(gq + W, real_lo, (size_t)W * 8);
        memcpy(gq + 2 * W, real_hi, (size_t)W * 8);
        memcpy(gq + 3 * W, starts.data(), (size_t)W * 8);
        int64_t *ps = gq + 4 * W;
        ps[0] = 0;
        for (int64_t w = 0; w < W; ++w) ps[w + 1] = ps[w] + (n_valid[w] + 1) / 2;
        const int64_t NP = ps[W];
        int32_t *g32 = reinterpret_cast<int32_t *>(gq + 5 * W + 1);
        memcpy(g32, rows, (size_t)W * 4);
        memcpy(g32 + W, n_valid, (size_t)W * 4);
        put(off_bgd, bgd, (size_t)bank->D * 4);
        put(off_starts, starts.data(), (size_t)W * 8);
        put(off_ends, ends.data(), (size_t)W * 8);
        const int64_t tiles_bound = (NP + (int64_t)15 * std::min<int64_t>(NB, NP)) / 16 + 1;
        void *d_wb = nullptr, *d_tiles = nullptr, *d_work = nullptr;
        TSR_TRY(tsr_scratch(ctx, SLOT_W, up.size(), &d_up));
        TSR_TRY(tsr_scratch(ctx, SLOT_RAW, (size_t)(bank->wm_total4 + 128) * 16, &d_wb));
        TSR_TRY(tsr_scratch(ctx, SLOT_AUX, (size_t)tiles_bound * 16 * sizeof(tsr_iso_pair), &d_tiles));
        TSR_TRY(tsr_scratch(ctx, SLOT_IN, ((size_t)3 * NB + 4 + (size_t)2 * tiles_bound) * 4, &d_work));
        TSR_HIP(hipMemcpyAsync(d_up, up.data(), up.size(), hipMemcpyHostToDevice, ctx->stream));
        const uint8_t *du = static_cast<const uint8_t *>(d_up);
        const int32_t *d_kr = nullptr;
        const int *d_nt = nullptr;
        TSR_TRY(tsr_launch_isolated_tables(ctx, (const int64_t *)(du + off_geo), W, NP, RMAX, (int *)d_work, (tsr_iso_pair *)d_tiles, tiles_bound,
                                           &d_kr, &d_nt));
        TSR_TRY(tsr_launch_isolated_mfma(ctx, bank, cp->d_feats, cp->Dp, (const float *)(du + off_bgd), (const tsr_iso_pair *)d_tiles, d_kr,
                                         tiles_bound, NW, d_wb, (float *)d_all, d_nt));
    } else {
        const size_t off_left = off_tab + (size_t)NW * 24;
        up.resize(off_left + (size_t)NW * 4 + 16);
        int64_t *it = reinterpret_cast<int64_t *>(up.data() + off_tab);
        int32_t *left = reinterpret_cast<int32_t *>(up.data() + off_left);
        for (int64_t w = 0; w < W; ++w)
            for (int32_t t = 0; t < n_valid[w]; ++t) {
                const size_t i = (size_t)(starts[(size_t)w] + t);
                it[i] = first[w] + t;
                it[(size_t)NW + i] = real_lo[w];
                it[(size_t)2 * NW + i] = real_hi[w];
                left[i] = rows[w] - t;
            }
        put(off_bgd, bgd, (size_t)bank->D * 4);
        put(off_starts, starts.data(), (size_t)W * 8);
        put(off_ends, ends.data(), (size_t)W * 8);
        TSR_TRY(tsr_scratch(ctx, SLOT_W, up.size(), &d_up));
        TSR_HIP(hipMemcpyAsync(d_up, up.data(), up.size(), hipMemcpyHostToDevice, ctx->stream));
        const uint8_t *du = static_cast<const uint8_t *>(d_up);
        const int64_t *d_first = (const int64_t *)(du + off_tab);
        TSR_TRY(tsr_launch_isolated_windows(ctx, bank, cp->d_feats, cp->Dp, (const float *)(du + off_bgd), d_first, d_first + NW,
                                            d_first + 2 * NW, (const int32_t *)(du + off_left), NW, (float *)d_all));
    }
    const uint8_t *du = static_cast<const uint8_t *>(d_up);
    TSR_TRY(tsr_launch_window_argmax(ctx, (const float *)d_all, NW, (int)bank->M, (const int64_t *)(du + off_starts),
                                     (const int64_t *)(du + off_ends), W, d_score, d_loc, d_tid));
    std::vector<uint8_t> res((size_t)W * 16);   // [loc W x 8][score W x 4][template W x 4], as the kernel leaves them
    TSR_HIP(hipMemcpyAsync(res.data(), d_res, res.size(), hipMemcpyDeviceToHost, ctx->stream));
    TSR_HIP(hipStreamSynchronize(ctx->stream));
    const int64_t *loc64 = reinterpret_cast<const int64_t *>(res.data());
    memcpy(score, res.data() + (size_t)W * 8, (size_t)W * 4);
    memcpy(template_id, res.data() + (size_t)W * 12, (size_t)W * 4);
    for (int64_t w = 0; w < W; ++w) loc[w] = (int32_t)(loc64[w] - starts[(size_t)w]);
    return TSR_OK;
}

// ---- peak picking on score rows ------------------------------------
// Both take either host score rows (`vals` f32 [offsets[N]], utterance n = [offsets[n], offsets[n+1]))
// or, with vals == NULL, the scores a previous tsr_corpus_score left on the device.
static int peaks_source(tsr_ctx *ctx, const tsr_corpus *cp, const float *vals, const int64_t *offsets, int64_t N,
                        const float **d_v, const int64_t **d_off, int64_t *total, const char *who)
{
    TSR_REQUIRE(ctx && (cp || (vals && offsets)), "%s: need a scored corpus or host score rows", who);
    TSR_REQUIRE(!cp || cp->scored || cp->P == 0, "%s: this corpus has not been scored (tsr_corpus_score) -- pass host score rows or score it first", who);
    TSR_HIP(hipSetDevice(ctx->device));
    const int64_t *offs = cp ? cp->offsets.data() : offsets;
    const int64_t n = cp ? cp->N : N;
    TSR_REQUIRE(n >= 0 && offs[0] == 0, "%s: bad offsets", who);
    for (int64_t i = 0; i < n; ++i) TSR_REQUIRE(offs[i + 1] >= offs[i], "%s: offsets must not decrease", who);
    *total = offs[n];
    void *dO = nullptr;
    TSR_TRY(tsr_scratch(ctx, SLOT_AUX, (size_t)(n + 1) * 8, &dO));
    TSR_HIP(hipMemcpyAsync(dO, offs, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    *d_off = (const int64_t *)dO;
    if (cp) {
        *d_v = cp->d_scores;
    } else {
        void *dV = nullptr;
        TSR_TRY(tsr_scratch(ctx, SLOT_IN, (size_t)std::max<int64_t>(*total, 1) * 4, &dV));
        if (*total) TSR_HIP(hipMemcpyAsync(dV, vals, (size_t)*total * 4, hipMemcpyHostToDevice, ctx->stream));
        *d_v = (const float *)dV;
    }
    return TSR_OK;
}

TSR_API int tsr_peaks(tsr_ctx *ctx, const tsr_corpus *cp, const float *vals, const int64_t *offsets, int64_t N,
                      float threshold, int C0, int64_t *counts, int32_t *times)
{
    TSR_REQUIRE(counts && times, "tsr_peaks: null output");
    const float *d_v;
    const int64_t *d_off;
    int64_t total;
    TSR_TRY(peaks_source(ctx, cp, vals, offsets, N, &d_v, &d_off, &total, "tsr_peaks"));
    const int64_t n = cp ? cp->N : N;
    if (n == 0) return TSR_OK;
    void *dC = nullptr, *dT = nullptr;
    TSR_TRY(tsr_scratch(ctx, SLOT_AUX2, (size_t)n * 8, &dC));
    TSR_TRY(tsr_scratch(ctx, SLOT_W, (size_t)std::max<int64_t>(total, 1) * 4, &dT));
    TSR_TRY(tsr_launch_peaks(ctx, d_v, d_off, n, threshold, C0, (int64_t *)dC, (int32_t *)dT));
    TSR_HIP(hipMemcpyAsync(counts, dC, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (total) TSR_TRY(tsr_download(ctx, times, dT, (size_t)total * 4));
    TSR_HIP(hipStreamSynchronize(ctx->stream));
    return TSR_OK;
}

// _get_detect_clusters_single_threshold (roc_functions.py:356-383) for every utterance: peaks above the
// threshold (tsr_peaks), clustered by cluster_times.pyx:133-222, as (start, end) pairs.  peak_times may be
// NULL; when given it also receives the peak lists (as tsr_peaks stores them) with their counts.
TSR_API int tsr_detect_clusters(tsr_ctx *ctx, const tsr_corpus *cp, const float *vals, const int64_t *offsets, int64_t N,
                                float threshold, float C0, float C1, int64_t *n_clusters, int32_t *starts, int32_t *ends,
                                int64_t *peak_counts, int32_t *peak_times)
{
    TSR_REQUIRE(n_clusters && starts && ends, "tsr_detect_clusters: null output");
    TSR_REQUIRE(C0 >= 1.0f && C0 == (float)(int)C0, "tsr_detect_clusters: C0 = %g must be a whole number >= 1 (it also bounds the peak search, roc_functions.py:178)", (double)C0);
    const float *d_v;
    const int64_t *d_off;
    int64_t total;
    TSR_TRY(peaks_source(ctx, cp, vals, offsets, N, &d_v, &d_off, &total, "tsr_detect_clusters"));
    const int64_t n = cp ? cp->N : N;
    const int64_t *offs = cp ? cp->offsets.data() : offsets;
    if (n == 0) return TSR_OK;
    for (int64_t i = 0; i < n; ++i)
        TSR_REQUIRE(offs[i + 1] - offs[i] < 65535, "tsr_detect_clusters: utterance %lld has %lld scores; the reference keeps peak times in uint16", (long long)i, (long long)(offs[i + 1] - offs[i]));
    void *dC = nullptr, *dT = nullptr, *dP = nullptr, *dR = nullptr;
    TSR_TRY(tsr_scratch(ctx, SLOT_AUX2, (size_t)n * 16, &dC));
    TSR_TRY(tsr_scratch(ctx, SLOT_W, (size_t)std::max<int64_t>(total, 1) * 4, &dT));
    TSR_TRY(tsr_scratch(ctx, SLOT_RAW, (size_t)(total + n + 16), &dP));
    TSR_TRY(tsr_scratch(ctx, SLOT_OUT, (size_t)std::max<int64_t>(total, 1) * 8, &dR));
    int64_t *d_counts = (int64_t *)dC, *d_nc = d_counts + n;
    int32_t *d_starts = (int32_t *)dR, *d_ends = d_starts + std::max<int64_t>(total, 1);
    TSR_TRY(tsr_launch_peaks(ctx, d_v, d_off, n, threshold, (int)C0, d_counts, (int32_t *)dT));
    TSR_TRY(tsr_launch_cluster_times(ctx, (const int32_t *)dT, d_off, d_counts, n, C0, C1, nullptr, (uint8_t *)dP, d_nc, d_starts, d_ends));
    TSR_HIP(hipMemcpyAsync(n_clusters, d_nc, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (peak_counts) TSR_HIP(hipMemcpyAsync(peak_counts, d_counts, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (total) {
        TSR_TRY(tsr_download(ctx, starts, d_starts, (size_t)total * 4));
        TSR_TRY(tsr_download(ctx, ends, d_ends, (size_t)total * 4));
        if (peak_times) TSR_TRY(tsr_download(ctx, peak_times, dT, (size_t)total * 4));
    }
    TSR_HIP(hipStreamSynchronize(ctx->stream));
    return TSR_OK;
}

// One list of ascending peak times through cluster_times_kernel: the partition codes (n + 1 bytes).
// lens32 (may be null): the template length of every detection (cluster_times_template_lengths).
static int cluster_one_list(tsr_ctx *ctx, const uint16_t *times, const int32_t *lens32, int64_t n, float C0, float C1,
                            uint8_t *partition, const char *who)
{
    TSR_REQUIRE(n >= 0 && n < 65535, "%s: %lld times (the reference counts them in uint16)", who, (long long)n);
    for (int64_t i = 1; i < n; ++i)
        TSR_REQUIRE(times[i] > times[i - 1], "%s: times must be strictly increasing (index %lld)", who, (long long)i);
    TSR_HIP(hipSetDevice(ctx->device));
    const size_t m = (size_t)std::max<int64_t>(n, 1);
    std::vector<int32_t> t32(2 * m, 0);  // times, then lengths
    for (int64_t i = 0; i < n; ++i) t32[(size_t)i] = times[i];
    if (lens32)
        for (int64_t i = 0; i < n; ++i) t32[m + (size_t)i] = lens32[i];
    const int64_t offs[2] = {0, n}, cnt = n;
    void *dT = nullptr, *dM = nullptr, *dP = nullptr, *dR = nullptr;
    TSR_TRY(tsr_scratch(ctx, SLOT_W, t32.size() * 4, &dT));
    TSR_TRY(tsr_scratch(ctx, SLOT_AUX, 64, &dM));
    TSR_TRY(tsr_scratch(ctx, SLOT_RAW, (size_t)n + 16, &dP));
    TSR_TRY(tsr_scratch(ctx, SLOT_OUT, m * 8, &dR));
    int64_t *d_off = (int64_t *)dM, *d_cnt = d_off + 2, *d_nc = d_off + 3;
    TSR_HIP(hipMemcpyAsync(dT, t32.data(), t32.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    TSR_HIP(hipMemcpyAsync(d_off, offs, 16, hipMemcpyHostToDevice, ctx->stream));
    TSR_HIP(hipMemcpyAsync(d_cnt, &cnt, 8, hipMemcpyHostToDevice, ctx->stream));
    TSR_TRY(tsr_launch_cluster_times(ctx, (const int32_t *)dT, d_off, d_cnt, 1, C0, C1,
                                     lens32 ? (const int32_t *)dT + m : nullptr, (uint8_t *)dP, d_nc, (int32_t *)dR,
                                     (int32_t *)dR + m));
    TSR_HIP(hipMemcpyAsync(partition, dP, (size_t)n + 1, hipMemcpyDeviceToHost, ctx->stream));
    TSR_HIP(hipStreamSynchronize(ctx->stream));
    return TSR_OK;
}

// cluster_times.pyx:133-222 for ONE list of ascending peak times.
TSR_API int tsr_cluster_times(tsr_ctx *ctx, const uint16_t *times, int64_t n, float C0, float C1, uint8_t *partition)
{
    TSR_REQUIRE(ctx && partition && (times || n == 0), "tsr_cluster_times: null pointer");
    return cluster_one_list(ctx, times, nullptr, n, C0, C1, partition, "tsr_cluster_times");
}

// cluster_times.pyx:221-310 (cluster_times_template_lengths) for ONE list of ascending peak times.
TSR_API int tsr_cluster_times_template_lengths(tsr_ctx *ctx, const uint16_t *times, const uint16_t *template_lengths,
                                               int64_t n_lengths, const uint16_t *template_components, int64_t n,
                                               uint8_t *partition)
{
    TSR_REQUIRE(ctx && partition && ((times && template_components && template_lengths) || n == 0),
                "tsr_cluster_times_template_lengths: null pointer");
    TSR_REQUIRE(n >= 0 && n_lengths >= 0, "tsr_cluster_times_template_lengths: negative size");
    std::vector<int32_t> lens((size_t)std::max<int64_t>(n, 1), 0);
    for (int64_t i = 0; i < n; ++i) {
        TSR_REQUIRE(template_components[i] < n_lengths,
                    "tsr_cluster_times_template_lengths: template_components[%lld] = %d but only %lld template lengths "
                    "(the reference reads out of bounds here)",
                    (long long)i, (int)template_components[i], (long long)n_lengths);
        lens[(size_t)i] = template_lengths[template_components[i]];
    }
    return cluster_one_list(ctx, times, lens.data(), n, 0.0f, 0.0f, partition, "tsr_cluster_times_template_lengths");
}

// cluster_times.pyx:73-131 (cluster_times_variable_length) for ONE list of detection times: partition u8 [n].
TSR_API int tsr_cluster_times_variable_length(tsr_ctx *ctx, const int32_t *times, const int32_t *detection_lengths,
                                              int64_t n, uint8_t *partition)
{
    TSR_REQUIRE(ctx && ((times && detection_lengths && partition) || n == 0),
                "tsr_cluster_times_variable_length: null pointer");
    TSR_REQUIRE(n >= 0, "tsr_cluster_times_variable_length: negative size");
    if (n == 0) return TSR_OK;
    TSR_HIP(hipSetDevice(ctx->device));
    void *dT = nullptr, *dP = nullptr;
    TSR_TRY(tsr_scratch(ctx, SLOT_W, (size_t)n * 8, &dT));
    TSR_TRY(tsr_scratch(ctx, SLOT_RAW, (size_t)n + 16, &dP));
    int32_t *d_t = (int32_t *)dT, *d_l = d_t + n;
    TSR_HIP(hipMemcpyAsync(d_t, times, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    TSR_HIP(hipMemcpyAsync(d_l, detection_lengths, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    TSR_TRY(tsr_launch_link_partition(ctx, d_t, d_l, n, (uint8_t *)dP));
    TSR_HIP(hipMemcpyAsync(partition, dP, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    TSR_HIP(hipStreamSynchronize(ctx->stream));
    return TSR_OK;
}

TSR_API int tsr_window_peak_max(tsr_ctx *ctx, const tsr_corpus *cp, const float *vals, const int64_t *offsets,
                                int64_t N, const int64_t *utt, const int64_t *starts, const int64_t *ends, int64_t W,
                                float *out)
{
    TSR_REQUIRE(W >= 0 && (W == 0 || (utt && starts && ends && out)), "tsr_window_peak_max: null pointer");
    const float *d_v;
    const int64_t *d_off;
    int64_t total;
    TSR_TRY(peaks_source(ctx, cp, vals, offsets, N, &d_v, &d_off, &total, "tsr_window_peak_max"));
    const int64_t n = cp ? cp->N : N;
    if (W == 0) return TSR_OK;
    for (int64_t w = 0; w < W; ++w)
        TSR_REQUIRE(utt[w] >= 0 && utt[w] < n && starts[w] >= 0, "tsr_window_peak_max: window %lld refers to utterance %lld / start %lld", (long long)w, (long long)utt[w], (long long)starts[w]);
    void *dWn = nullptr, *dR = nullptr;
    TSR_TRY(tsr_scratch(ctx, SLOT_AUX2, (size_t)W * 24, &dWn));
    TSR_TRY(tsr_scratch(ctx, SLOT_W, (size_t)W * 4, &dR));
    int64_t *d_utt = (int64_t *)dWn, *d_s = d_utt + W, *d_e = d_s + W;
    TSR_HIP(hipMemcpyAsync(d_utt, utt, (size_t)W * 8, hipMemcpyHostToDevice, ctx->stream));
    TSR_HIP(hipMemcpyAsync(d_s, starts, (size_t)W * 8, hipMemcpyHostToDevice, ctx->stream));
    TSR_HIP(hipMemcpyAsync(d_e, ends, (size_t)W * 8, hipMemcpyHostToDevice, ctx->stream));
    TSR_TRY(tsr_launch_window_peak_max(ctx, d_v, d_off, d_utt, d_s, d_e, W, (float *)dR));
    TSR_HIP(hipMemcpyAsync(out, dR, (size_t)W * 4, hipMemcpyDeviceToHost, ctx->stream));
    TSR_HIP(hipStreamSynchronize(ctx->stream));
    return TSR_OK;
}

// ---- edge front-end ---------------------------------------------------
// get_edge_features, default branch (get_train_data.py:2547-2557): spectrograms -> a resident binary
// edge corpus (D = 8 H, layout [frame][h][edge type] = reorg_part_for_fast_filtering).
static int edge_corpus_create_impl(tsr_ctx *ctx, const double *S_cat, const uint8_t *mask_cat, const int64_t *frame_offsets, int64_t N,
                                   int64_t H, int block_length, int spread_length, double quantile,
                                   const double *abst_threshold, tsr_corpus **out)
{
    TSR_REQUIRE(ctx && frame_offsets && out && abst_threshold, "tsr_edge_corpus_create: null pointer");
    *out = nullptr;
    TSR_REQUIRE(N >= 0 && H >= 1 && H <= 4096 && block_length >= 1 && spread_length >= 0,
                "tsr_edge_corpus_create: bad geometry (H=%lld, block_length=%d, spread_length=%d)", (long long)H, block_length, spread_length);
    TSR_REQUIRE(quantile >= 0.0 && quantile < 1.0, "tsr_edge_corpus_create: quantile %g outside [0, 1) (the reference indexes past its sorted values at 1)", quantile);
    TSR_REQUIRE(frame_offsets[0] == 0, "tsr_edge_corpus_create: frame_offsets[0] must be 0");
    std::vector<int32_t> items;
    for (int64_t n = 0; n < N; ++n) {
        const int64_t T = frame_offsets[n + 1] - frame_offsets[n];
        TSR_REQUIRE(T >= 0 && T < ((int64_t)1 << 30), "tsr_edge_corpus_create: bad utterance length");
        const int64_t full = T / block_length, last = T % block_length;
        if (last > 0) {  // the trailing partial block: thresholded on the last block_length columns, only its own kept
            const int64_t c0 = std::max<int64_t>(0, T - block_length), w = std::min<int64_t>(block_length, T);
            items.insert(items.end(), {(int32_t)n, (int32_t)c0, (int32_t)w, (int32_t)(w - last)});
        }
        for (int64_t b = 0; b < full; ++b) items.insert(items.end(), {(int32_t)n, (int32_t)(b * block_length), block_length, 0});
    }
    const int64_t P = frame_offsets[N];
    TSR_REQUIRE(P == 0 || S_cat, "tsr_edge_corpus_create: null spectrogram");
    TSR_HIP(hipSetDevice(ctx->device));
    tsr_corpus *cp = nullptr;
    TSR_TRY(tsr_corpus_create_empty(ctx, frame_offsets, N, 8 * H, &cp));
    void *dS = nullptr, *dO = nullptr, *dI = nullptr, *dM = nullptr;
    int s = tsr_scratch(ctx, SLOT_RAW, (size_t)std::max<int64_t>(P * H, 1) * 8, &dS);
    if (s == TSR_OK && mask_cat && P > 0) {
        s = tsr_scratch(ctx, SLOT_IN, (size_t)(P * H), &dM);
        if (s == TSR_OK && hipMemcpyAsync(dM, mask_cat, (size_t)(P * H), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) s = TSR_ERR_HIP;
    }
    if (s == TSR_OK) s = tsr_scratch(ctx, SLOT_AUX, (size_t)(N + 1) * 8, &dO);
    if (s == TSR_OK) s = tsr_scratch(ctx, SLOT_W, std::max<size_t>(items.size(), 4) * 4, &dI);
    if (s == TSR_OK && P > 0 &&
        (hipMemcpyAsync(dS, S_cat, (size_t)(P * H) * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
         hipMemcpyAsync(dO, frame_offsets, (size_t)(N + 1) * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
         hipMemcpyAsync(dI, items.data(), items.size() * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess))
        s = TSR_ERR_HIP;
    if (s == TSR_OK && P > 0)
        s = tsr_launch_edge_blocks(ctx, (const double *)dS, (const uint8_t *)dM, (const int64_t *)dO, (const int32_t *)dI, (int64_t)(items.size() / 4),
                                   (int)H, block_length, spread_length, quantile, abst_threshold, cp->Dp, cp->d_feats);
    if (s == TSR_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) s = TSR_ERR_HIP;
    if (s != TSR_OK) {
        if (s == TSR_ERR_HIP) tsr_set_error("tsr_edge_corpus_create: device copy/launch failed");
        if (s == TSR_ERR_UNSUPPORTED) tsr_set_error("tsr_edge_corpus_create: H=%lld x block_length=%d does not fit the LDS", (long long)H, block_length);
        tsr_corpus_destroy(cp);
        return s;
    }
    *out = cp;
    return TSR_OK;
}

TSR_API int tsr_edge_corpus_create(tsr_ctx *ctx, const double *S_cat, const int64_t *frame_offsets, int64_t N,
                                   int64_t H, int block_length, int spread_length, double quantile,
                                   const double *abst_threshold, tsr_corpus **out)
{
    return edge_corpus_create_impl(ctx, S_cat, nullptr, frame_offsets, N, H, block_length, spread_length, quantile, abst_threshold, out);
}

TSR_API int tsr_edge_corpus_create_masked(tsr_ctx *ctx, const double *S_cat, const uint8_t *mask_cat, const int64_t *frame_offsets,
                                          int64_t N, int64_t H, int block_length, int spread_length, double quantile,
                                          const double *abst_threshold, tsr_corpus **out)
{
    TSR_REQUIRE(mask_cat || (frame_offsets && frame_offsets[N] == 0), "tsr_edge_corpus_create_masked: null mask");
    return edge_corpus_create_impl(ctx, S_cat, mask_cat, frame_offsets, N, H, block_length, spread_length, quantile, abst_threshold, out);
}

// ---- whole-array thresholding / spreading of an edge-strength map (edge_signal_proc.py:1184-1307, :1397-1417) ----
TSR_API int tsr_threshold_edgemap(tsr_ctx *ctx, double *E, const int64_t *row_breaks, int32_t n_blocks, int64_t T,
                                  double quantile, const double *abst_threshold, int32_t convert_to_binary, double *thresholds)
{
    TSR_REQUIRE(ctx && E && row_breaks && abst_threshold, "tsr_threshold_edgemap: null pointer");
    TSR_REQUIRE(n_blocks >= 0 && n_blocks <= 4096 && T >= 0, "tsr_threshold_edgemap: bad geometry");
    TSR_REQUIRE(quantile >= 0.0 && quantile < 1.0, "tsr_threshold_edgemap: quantile %g outside [0, 1) (the reference indexes past its sorted values at 1)", quantile);
    for (int b = 0; b < n_blocks; ++b)
        TSR_REQUIRE(row_breaks[b] >= 0 && row_breaks[b + 1] >= row_breaks[b], "tsr_threshold_edgemap: row breaks must ascend");
    if (n_blocks == 0 || T == 0) return TSR_OK;
    const int64_t rows = row_breaks[n_blocks];
    TSR_HIP(hipSetDevice(ctx->device));
    void *dE = nullptr, *dB = nullptr, *dA = nullptr;
    TSR_TRY(tsr_scratch(ctx, SLOT_RAW, (size_t)std::max<int64_t>(rows * T, 1) * 8, &dE));
    TSR_TRY(tsr_scratch(ctx, SLOT_AUX, (size_t)(n_blocks + 1) * 8, &dB));
    TSR_TRY(tsr_scratch(ctx, SLOT_AUX2, (size_t)n_blocks * 16, &dA));  // [abst | taus]
    TSR_HIP(hipMemcpyAsync(dE, E, (size_t)(rows * T) * 8, hipMemcpyHostToDevice, ctx->stream));
    TSR_HIP(hipMemcpyAsync(dB, row_breaks, (size_t)(n_blocks + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    TSR_HIP(hipMemcpyAsync(dA, abst_threshold, (size_t)n_blocks * 8, hipMemcpyHostToDevice, ctx->stream));
    double *d_taus = (double *)dA + n_blocks;
    TSR_TRY(tsr_launch_threshold_edgemap(ctx, (double *)dE, (const int64_t *)dB, n_blocks, T, quantile, (const double *)dA,
                                         convert_to_binary ? 1 : 0, d_taus));
    // only the rows inside [breaks[0], breaks[n]) are touched
    const int64_t r0 = row_breaks[0];
    TSR_HIP(hipMemcpyAsync(E + r0 * T, (double *)dE + r0 * T, (size_t)((rows - r0) * T) * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (thresholds) TSR_HIP(hipMemcpyAsync(thresholds, d_taus, (size_t)n_blocks * 8, hipMemcpyDeviceToHost, ctx->stream));
    TSR_HIP(hipStreamSynchronize(ctx->stream));
    return TSR_OK;
}

TSR_API int tsr_spread_edgemap(tsr_ctx *ctx, double *E, const int64_t *row_breaks, int32_t n_blocks, int64_t T,
                               const int32_t *orientations, int32_t spread_length)
{
    TSR_REQUIRE(ctx && E && row_breaks, "tsr_spread_edgemap: null pointer");
    TSR_REQUIRE(n_blocks >= 0 && n_blocks <= 4096 && T >= 0, "tsr_spread_edgemap: bad geometry");
    TSR_REQUIRE(spread_length >= 1, "tsr_spread_edgemap: spread_length must be >= 1 (the reference's zero-padded copy is empty at 0)");
    for (int b = 0; b < n_blocks; ++b)
        TSR_REQUIRE(row_breaks[b] >= 0 && row_breaks[b + 1] >= row_breaks[b], "tsr_spread_edgemap: row breaks must ascend");
    if (n_blocks == 0 || T == 0) return TSR_OK;
    const int64_t rows = row_breaks[n_blocks];
    TSR_HIP(hipSetDevice(ctx->device));
    void *dE = nullptr, *dO = nullptr, *dB = nullptr, *dD = nullptr;
    TSR_TRY(tsr_scratch(ctx, SLOT_RAW, (size_t)std::max<int64_t>(rows * T, 1) * 8, &dE));
    TSR_TRY(tsr_scratch(ctx, SLOT_OUT, (size_t)std::max<int64_t>(rows * T, 1) * 8, &dO));
    TSR_TRY(tsr_scratch(ctx, SLOT_AUX, (size_t)(n_blocks + 1) * 8, &dB));
    TSR_TRY(tsr_scratch(ctx, SLOT_AUX2, (size_t)n_blocks * 8, &dD));
    TSR_HIP(hipMemcpyAsync(dE, E, (size_t)(rows * T) * 8, hipMemcpyHostToDevice, ctx->stream));
    TSR_HIP(hipMemcpyAsync(dB, row_breaks, (size_t)(n_blocks + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    if (orientations) TSR_HIP(hipMemcpyAsync(dD, orientations, (size_t)n_blocks * 8, hipMemcpyHostToDevice, ctx->stream));
    TSR_TRY(tsr_launch_spread_edgemap(ctx, (const double *)dE, (double *)dO, (const int64_t *)dB, n_blocks, rows, T,
                                      orientations ? (const int32_t *)dD : nullptr, spread_length));
    const int64_t r0 = row_breaks[0];
    TSR_HIP(hipMemcpyAsync(E + r0 * T, (double *)dO + r0 * T, (size_t)((rows - r0) * T) * 8, hipMemcpyDeviceToHost, ctx->stream));
    TSR_HIP(hipStreamSynchronize(ctx->stream));
    return TSR_OK;
}

// ---- magnitude features (edge_signal_proc.py:600-747) ------------------------------------------------------------
namespace {
struct MagWinHost {  // mirrors edges2.hip's MagWin
    int a, b, s0, s1;
    long long k;
};
}  // namespace

TSR_API int tsr_magnitude_features(tsr_ctx *ctx, const double *S, int64_t F0, int64_t T, int32_t block_length,
                                   int32_t spread_radius, double quantile, int32_t mag_smooth_freq, int32_t mag_downsample_freq,
                                   int32_t mag_spread_length, int32_t whole_block, uint8_t *out)
{
    TSR_REQUIRE(ctx && S && out, "tsr_magnitude_features: null pointer");
    TSR_REQUIRE(F0 >= 1 && F0 <= 65536 && T >= 1 && T < ((int64_t)1 << 30), "tsr_magnitude_features: bad shape [%lld, %lld]", (long long)F0, (long long)T);
    TSR_REQUIRE(block_length >= 4, "tsr_magnitude_features: block_length %d < 4 (the reference's window never advances)", block_length);
    TSR_REQUIRE(mag_smooth_freq >= 0 && mag_downsample_freq >= 0 && mag_spread_length >= 0 && spread_radius >= 0 && spread_radius <= 64,
                "tsr_magnitude_features: negative or oversized parameter");
    TSR_REQUIRE(quantile >= 0.0, "tsr_magnitude_features: negative quantile");
    const int down = mag_downsample_freq;
    const int64_t F = down > 0 ? (F0 + down - 1) / down : F0;
    const int bq = block_length / 4;
    // the estimation windows and the frames they set (:648-666 / :720-738); ranks beyond a window's sorted
    // values are an IndexError in the reference
    std::vector<MagWinHost> wins;
    const int64_t per = whole_block ? F : 1;
    {
        MagWinHost w;
        w.a = 0;
        w.b = (int)std::min<int64_t>(3 * bq, T);
        w.s0 = 0;
        w.s1 = (int)std::min<int64_t>(2 * bq, T);
        w.k = (long long)(quantile * 3 * bq * (double)per + .5);
        wins.push_back(w);
        int64_t next = 2 * bq;
        while (next < T) {
            w.a = (int)(next - bq);
            w.b = (int)std::min<int64_t>(T, (int64_t)w.a + 4 * bq);
            w.s0 = (int)next;
            w.s1 = (int)std::min<int64_t>(next + 2 * bq, T);
            w.k = (long long)(quantile * (double)(w.b - w.a) * (double)per);
            wins.push_back(w);
            next = w.s1;
        }
    }
    for (const MagWinHost &w : wins)
        TSR_REQUIRE(w.k >= 0 && w.k < (long long)(w.b - w.a) * per,
                    "tsr_magnitude_features: rank %lld outside a window of %lld values (IndexError in the reference)", w.k,
                    (long long)(w.b - w.a) * per);
    int min_count = 0;
    if (whole_block && spread_radius > 0) {  // correlate(E, w) >= 1 with E uint8: the n-fold repeated sum of fl(1/r) reaches 1.0
        const double x = 1.0 / spread_radius;
        double acc = 0.0;
        while (acc < 1.0) {
            acc += x;
            ++min_count;
        }
    }
    TSR_HIP(hipSetDevice(ctx->device));
    void *dS = nullptr, *dS2 = nullptr, *dW = nullptr, *dE = nullptr, *dO = nullptr;
    TSR_TRY(tsr_scratch(ctx, SLOT_RAW, (size_t)(F0 * T) * 8, &dS));
    TSR_TRY(tsr_scratch(ctx, SLOT_OUT, (size_t)(F * T) * 8, &dS2));
    TSR_TRY(tsr_scratch(ctx, SLOT_AUX, wins.size() * sizeof(MagWinHost), &dW));
    TSR_TRY(tsr_scratch(ctx, SLOT_AUX2, (size_t)(F * T), &dE));
    TSR_TRY(tsr_scratch(ctx, SLOT_IN, (size_t)(F * T), &dO));
    TSR_HIP(hipMemcpyAsync(dS, S, (size_t)(F0 * T) * 8, hipMemcpyHostToDevice, ctx->stream));
    TSR_HIP(hipMemcpyAsync(dW, wins.data(), wins.size() * sizeof(MagWinHost), hipMemcpyHostToDevice, ctx->stream));
    TSR_TRY(tsr_launch_magnitude(ctx, (const double *)dS, (int)F0, T, mag_smooth_freq, down, (int)F, (double *)dS2, dW, (int)wins.size(),
                                 whole_block ? 1 : 0, mag_spread_length, spread_radius, min_count, (uint8_t *)dE, (uint8_t *)dO));
    TSR_HIP(hipMemcpyAsync(out, dO, (size_t)(F * T), hipMemcpyDeviceToHost, ctx->stream));
    TSR_HIP(hipStreamSynchronize(ctx->stream));
    return TSR_OK;
}

// ---- Bernoulli-mixture M-step sums --------------------------------------
TSR_API int tsr_corpus_weighted_sums(tsr_ctx *ctx, const tsr_corpus *cp, const float *weights, int64_t K, float *out)
{
    TSR_REQUIRE(ctx && cp && weights && out, "tsr_corpus_weighted_sums: null pointer");
    TSR_REQUIRE(K >= 1 && K <= 4096, "tsr_corpus_weighted_sums: K=%lld out of range", (long long)K);
    const int64_t P = cp->P, D = cp->D;
    if (P == 0) {
        memset(out, 0, (size_t)(K * D) * sizeof(float));
        return TSR_OK;
    }
    TSR_HIP(hipSetDevice(ctx->device));
    const int S = (int)std::max<int64_t>(1, std::min<int64_t>(64, P / 64));  // frame slices (fixed reduction order)
    void *dA = nullptr, *dP = nullptr, *dO = nullptr;
    TSR_TRY(tsr_scratch(ctx, SLOT_IN, (size_t)(P * K) * 4, &dA));
    TSR_TRY(tsr_scratch(ctx, SLOT_RAW, (size_t)S * K * D * 4, &dP));
    TSR_TRY(tsr_scratch(ctx, SLOT_OUT, (size_t)(K * D) * 4, &dO));
    TSR_HIP(hipMemcpyAsync(dA, weights, (size_t)(P * K) * 4, hipMemcpyHostToDevice, ctx->stream));
    TSR_TRY(tsr_corpus_need_bytes(ctx, cp));
    TSR_TRY(tsr_launch_mstep(ctx, cp->d_feats, P, (int)D, cp->Dp, (const float *)dA, (int)K, (float *)dP, S, (float *)dO));
    TSR_HIP(hipMemcpyAsync(out, dO, (size_t)(K * D) * 4, hipMemcpyDeviceToHost, ctx->stream));
    TSR_HIP(hipStreamSynchronize(ctx->stream));
    return TSR_OK;
}

// ---- Bernoulli-mixture EM on a resident data matrix (bernoulli_mixture.py:108-291) --------
// Component scores log P(x_n | k) = consts[k] + sum_d x[n,d] log_odds[k,d] of the rows of a resident data matrix
// into a device array [K][P]: the tables go up zero padded to the resident row stride, one wave per row multiplies
// (classify.hip: mixture_scores_kernel).
static int mixture_scores_device(tsr_ctx *ctx, const tsr_corpus *cp, const float *log_odds, const double *consts, int K, float *d_all)
{
    const int64_t D = cp->D;
    const int Dq = cp->Dp;
    std::vector<float> tab((size_t)K * Dq, 0.0f);
    for (int k = 0; k < K; ++k) {
        for (int64_t d = 0; d < D; ++d) TSR_REQUIRE(std::isfinite(log_odds[(size_t)k * D + d]), "mixture: log-odds value (%d, %lld) is not finite", k, (long long)d);
        TSR_REQUIRE(std::isfinite(consts[k]), "mixture: constant %d is not finite", k);
        memcpy(tab.data() + (size_t)k * Dq, log_odds + (size_t)k * D, (size_t)D * sizeof(float));
    }
    void *d_tab = nullptr, *d_c = nullptr;
    TSR_TRY(tsr_scratch(ctx, SLOT_W, tab.size() * sizeof(float), &d_tab));
    TSR_TRY(tsr_scratch(ctx, SLOT_AUX, (size_t)K * 8, &d_c));
    TSR_HIP(hipMemcpyAsync(d_tab, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    TSR_HIP(hipMemcpyAsync(d_c, consts, (size_t)K * 8, hipMemcpyHostToDevice, ctx->stream));
    TSR_TRY(tsr_corpus_need_bytes(ctx, cp));
    TSR_TRY(tsr_launch_mixture_scores(ctx, cp->d_feats, cp->P, (int)D, cp->Dp, (const float *)d_tab, Dq, (const double *)d_c, K, d_all));
    TSR_HIP(hipStreamSynchronize(ctx->stream));  // (the host staging vector dies here)
    return TSR_OK;
}

TSR_API int tsr_mixture_scores(tsr_ctx *ctx, const tsr_corpus *cp, const float *log_odds, const double *consts, int32_t K,
                               float *scores)
{
    TSR_REQUIRE(ctx && cp && log_odds && consts && (scores || cp->P == 0), "tsr_mixture_scores: null pointer");
    TSR_REQUIRE(K >= 1 && K <= 65535, "tsr_mixture_scores: K=%d out of range", K);
    if (cp->P == 0) return TSR_OK;
    TSR_HIP(hipSetDevice(ctx->device));
    void *d_all = nullptr;
    TSR_TRY(tsr_scratch(ctx, SLOT_OUT, (size_t)K * cp->P * sizeof(float), &d_all));
    TSR_TRY(mixture_scores_device(ctx, cp, log_odds, consts, K, (float *)d_all));
    return tsr_download(ctx, scores, d_all, (size_t)K * cp->P * sizeof(float));
}

// E-step: component scores, then log marginals and posteriors on the device; the posteriors stay there for the
// M-step.  log_odds[k] = log T_k - log(1 - T_k), consts[k] = sum_d log(1 - T_k[d]).
TSR_API int tsr_mixture_e_step(tsr_ctx *ctx, tsr_corpus *cp, const float *log_odds, const double *consts, int32_t K,
                               const double *log_weights, int32_t hard_assignment, double *loglikelihood, float *affinities)
{
    TSR_REQUIRE(ctx && cp && log_odds && consts && log_weights && loglikelihood, "tsr_mixture_e_step: null pointer");
    TSR_REQUIRE(K >= 1 && K <= 65535, "tsr_mixture_e_step: K=%d out of range", K);
    const int64_t P = cp->P;
    *loglikelihood = 0.0;
    // "an E-step has run" (aff_K >= 1) is committed only when everything below has succeeded: a failure in between must
    // not leave a later M-step or download reading posteriors that were never written (or lie in the old K layout)
    cp->aff_K = 0;
    cp->aff_colsum.clear();
    if (P == 0) {
        cp->aff_K = K;
        cp->aff_colsum.assign((size_t)K, 0.0);
        return TSR_OK;
    }
    TSR_HIP(hipSetDevice(ctx->device));
    if ((int64_t)P * K > cp->aff_cap) {
        TSR_HIP(hipStreamSynchronize(ctx->stream));
        dev_free(cp->d_aff);
        cp->d_aff = nullptr;
        cp->aff_cap = 0;
        TSR_HIP(hipMalloc(reinterpret_cast<void **>(&cp->d_aff), (size_t)P * K * sizeof(float)));
        cp->aff_cap = (int64_t)P * K;
    }
    void *d_all = nullptr, *d_lw = nullptr, *d_blk = nullptr;
    TSR_TRY(tsr_scratch(ctx, SLOT_OUT, (size_t)K * P * sizeof(float), &d_all));
    TSR_TRY(mixture_scores_device(ctx, cp, log_odds, consts, K, (float *)d_all));
    const int blocks = (int)((P + 255) / 256);
    TSR_TRY(tsr_scratch(ctx, SLOT_AUX, (size_t)K * 8, &d_lw));  // (the constants that lived here are consumed)
    TSR_TRY(tsr_scratch(ctx, SLOT_AUX2, (size_t)blocks * (K + 1) * 8, &d_blk));
    TSR_HIP(hipMemcpyAsync(d_lw, log_weights, (size_t)K * 8, hipMemcpyHostToDevice, ctx->stream));
    int nb = 0;
    TSR_TRY(tsr_launch_mixture_estep(ctx, (const float *)d_all, P, K, (const double *)d_lw, hard_assignment ? 1 : 0, cp->d_aff,
                                     (double *)d_blk, &nb));
    std::vector<double> blk((size_t)nb * (K + 1));
    TSR_HIP(hipMemcpyAsync(blk.data(), d_blk, blk.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    TSR_HIP(hipStreamSynchronize(ctx->stream));
    double ll = 0.0;
    std::vector<double> colsum((size_t)K, 0.0);
    for (int b = 0; b < nb; ++b) {  // block order: reproducible
        ll += blk[(size_t)b * (K + 1)];
        for (int k = 0; k < K; ++k) colsum[(size_t)k] += blk[(size_t)b * (K + 1) + 1 + k];
    }
    if (affinities) TSR_TRY(tsr_download(ctx, affinities, cp->d_aff, (size_t)P * K * sizeof(float)));
    *loglikelihood = ll;
    cp->aff_colsum.swap(colsum);
    cp->aff_K = K;
    return TSR_OK;
}

// M-step sums from the posteriors the latest tsr_mixture_e_step left on the device: weights[k] = mean_n a[n,k]
// (:175) and sums[k, d] = sum_n a[n,k] x[n,d] (:179; the caller divides and clips as the reference does).
TSR_API int tsr_mixture_m_step(tsr_ctx *ctx, tsr_corpus *cp, double *weights, float *sums)
{
    TSR_REQUIRE(ctx && cp && weights && sums, "tsr_mixture_m_step: null pointer");
    TSR_REQUIRE(cp->aff_K >= 1 && (cp->P == 0 || cp->d_aff), "tsr_mixture_m_step: no E-step has run on this data matrix");
    const int64_t P = cp->P, D = cp->D, K = cp->aff_K;
    for (int64_t k = 0; k < K; ++k) weights[k] = P ? cp->aff_colsum[(size_t)k] / (double)P : 0.0;
    if (P == 0) {
        memset(sums, 0, (size_t)(K * D) * sizeof(float));
        return TSR_OK;
    }
    TSR_HIP(hipSetDevice(ctx->device));
    const int S = (int)std::max<int64_t>(1, std::min<int64_t>(64, P / 64));
    void *dP = nullptr, *dO = nullptr;
    TSR_TRY(tsr_scratch(ctx, SLOT_RAW, (size_t)S * K * D * 4, &dP));
    TSR_TRY(tsr_scratch(ctx, SLOT_OUT, (size_t)(K * D) * 4, &dO));
    TSR_TRY(tsr_corpus_need_bytes(ctx, cp));
    TSR_TRY(tsr_launch_mstep(ctx, cp->d_feats, P, (int)D, cp->Dp, cp->d_aff, (int)K, (float *)dP, S, (float *)dO));
    TSR_HIP(hipMemcpyAsync(sums, dO, (size_t)(K * D) * 4, hipMemcpyDeviceToHost, ctx->stream));
    TSR_HIP(hipStreamSynchronize(ctx->stream));
    return TSR_OK;
}

TSR_API int tsr_corpus_download_affinities(tsr_ctx *ctx, const tsr_corpus *cp, float *affinities)
{
    TSR_REQUIRE(ctx && cp && affinities, "tsr_corpus_download_affinities: null pointer");
    TSR_REQUIRE(cp->aff_K >= 1 && (cp->P == 0 || cp->d_aff), "tsr_corpus_download_affinities: no E-step has run on this data matrix");
    if (cp->P == 0) return TSR_OK;
    TSR_HIP(hipSetDevice(ctx->device));
    return tsr_download(ctx, affinities, cp->d_aff, (size_t)cp->P * cp->aff_K * sizeof(float));
}

// ---- single-utterance drop-ins ----------------------------------------------------
TSR_API int tsr_detect_u8(tsr_ctx *ctx, const uint8_t *F, int64_t T, int64_t D, const float *LF,
                          int64_t L, float *out)
{
    TSR_REQUIRE(ctx && F && LF && out, "tsr_detect_u8: null pointer");
    TSR_REQUIRE(L >= 1 && T >= L, "tsr_detect_u8: need T >= L >= 1 (T=%lld, L=%lld); the reference reads out of bounds here", (long long)T, (long long)L);
    TSR_REQUIRE(D >= 1 && T <= 65535 * 64 && L * D < ((int64_t)1 << 31), "tsr_detect_u8: sizes out of range");
    for (int64_t i = 0; i < L * D; ++i)
        TSR_REQUIRE(isfinite(LF[i]), "tsr_detect_u8: filter value %lld is not finite", (long long)i);
    // the per-call path: caller's bytes and filter through the ctx scratch, one small launch
    TSR_HIP(hipSetDevice(ctx->device));
    void *dF, *dW, *dO;
    TSR_TRY(tsr_scratch(ctx, SLOT_IN, (size_t)(T * D) + 16, &dF));
    TSR_TRY(tsr_scratch(ctx, SLOT_W, (size_t)(L * D) * 4, &dW));
    TSR_TRY(tsr_scratch(ctx, SLOT_OUT, (size_t)T * 4, &dO));
    TSR_HIP(hipMemcpyAsync(dF, F, (size_t)(T * D), hipMemcpyHostToDevice, ctx->stream));
    TSR_HIP(hipMemcpyAsync(dW, LF, (size_t)(L * D) * 4, hipMemcpyHostToDevice, ctx->stream));
    TSR_TRY(tsr_launch_detect_u8_single(ctx, (const uint8_t *)dF, T, D, (const float *)dW, L, (float *)dO));
    TSR_HIP(hipMemcpyAsync(out, dO, (size_t)T * 4, hipMemcpyDeviceToHost, ctx->stream));
    TSR_HIP(hipStreamSynchronize(ctx->stream));
    return TSR_OK;
}

static int detect_f32_common(tsr_ctx *ctx, const float *F, int64_t T, int64_t D, const float *LF,
                             int64_t M, int64_t L, const float *cs, std::vector<float> &host_all)
{
    TSR_REQUIRE(L >= 1 && T >= L && M >= 1 && D >= 1, "detect_f32: need T >= L >= 1 (T=%lld, L=%lld)", (long long)T, (long long)L);
    TSR_REQUIRE(T <= 65535 * 256 && L * D < ((int64_t)1 << 31), "detect_f32: sizes out of range");
    TSR_HIP(hipSetDevice(ctx->device));
    void *dF, *dW, *dO, *dC = nullptr;
    TSR_TRY(tsr_scratch(ctx, SLOT_IN, (size_t)(T * D) * 4, &dF));
    TSR_TRY(tsr_scratch(ctx, SLOT_W, (size_t)(M * L * D) * 4, &dW));
    TSR_TRY(tsr_scratch(ctx, SLOT_OUT, (size_t)(M * T) * 4, &dO));
    TSR_HIP(hipMemcpyAsync(dF, F, (size_t)(T * D) * 4, hipMemcpyHostToDevice, ctx->stream));
    TSR_HIP(hipMemcpyAsync(dW, LF, (size_t)(M * L * D) * 4, hipMemcpyHostToDevice, ctx->stream));
    if (cs) {
        TSR_TRY(tsr_scratch(ctx, SLOT_AUX, (size_t)M * 4, &dC));
        TSR_HIP(hipMemcpyAsync(dC, cs, (size_t)M * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    TSR_TRY(tsr_launch_detect_f32(ctx, (const float *)dF, T, D, (const float *)dW, M, L, (const float *)dC, (float *)dO, nullptr));
    host_all.resize((size_t)(M * T));
    TSR_HIP(hipMemcpyAsync(host_all.data(), dO, (size_t)(M * T) * 4, hipMemcpyDeviceToHost, ctx->stream));
    TSR_HIP(hipStreamSynchronize(ctx->stream));
    return TSR_OK;
}

TSR_API int tsr_detect_f32(tsr_ctx *ctx, const float *F, int64_t T, int64_t D, const float *LF,
                           int64_t L, float *out)
{
    TSR_REQUIRE(ctx && F && LF && out, "tsr_detect_f32: null pointer");
    std::vector<float> all;
    TSR_TRY(detect_f32_common(ctx, F, T, D, LF, 1, L, nullptr, all));
    memcpy(out, all.data(), (size_t)T * 4);
    return TSR_OK;
}

TSR_API int tsr_detect_f32_max(tsr_ctx *ctx, const float *F, int64_t T, int64_t D, const float *LF,
                               int64_t M, int64_t L, const float *cs, float *out, uint16_t *ids)
{
    TSR_REQUIRE(ctx && F && LF && cs && out && ids, "tsr_detect_f32_max: null pointer");
    TSR_REQUIRE(M <= 65535, "tsr_detect_f32_max: M=%lld exceeds the uint16 id range", (long long)M);
    std::vector<float> all;
    TSR_TRY(detect_f32_common(ctx, F, T, D, LF, M, L, cs, all));
    // first maximum, strict '>' (compute_likelihood_linear_filter.pyx:270-277); the
    // tail t > T-L stays (0, 0) like the reference's zero-initialised outputs.
    for (int64_t t = 0; t < T; ++t) {
        if (t + L > T) {
            out[t] = 0.0f;
            ids[t] = 0;
            continue;
        }
        float best = all[(size_t)t];
        uint16_t bi = 0;
        for (int64_t m = 1; m < M; ++m) {
            const float v = all[(size_t)(m * T + t)];
            if (v > best) {
                best = v;
                bi = (uint16_t)m;
            }
        }
        out[t] = best;
        ids[t] = bi;
    }
    return TSR_OK;
}

TSR_API int tsr_max_detect_examples_u8(tsr_ctx *ctx, const uint8_t *F, int64_t N, int64_t T,
                                       int64_t D, const double *LF, int64_t L,
                                       const int64_t *lengths, double *out)
{
    TSR_REQUIRE(ctx && F && LF && lengths && out, "tsr_max_detect_examples_u8: null pointer");
    TSR_REQUIRE(N >= 0 && T >= 1 && D >= 1 && L >= 1, "tsr_max_detect_examples_u8: bad sizes");
    TSR_REQUIRE(N <= 65535 && L * D < ((int64_t)1 << 31), "tsr_max_detect_examples_u8: sizes out of range");
    for (int64_t n = 0; n < N; ++n)
        TSR_REQUIRE(lengths[n] >= 0 && lengths[n] <= T, "tsr_max_detect_examples_u8: lengths[%lld]=%lld outside [0,T]", (long long)n, (long long)lengths[n]);
    if (N == 0) return TSR_OK;
    TSR_HIP(hipSetDevice(ctx->device));
    void *dF, *dW, *dT;
    TSR_TRY(tsr_scratch(ctx, SLOT_IN, (size_t)(N * T * D), &dF));
    TSR_TRY(tsr_scratch(ctx, SLOT_W, (size_t)(L * D) * 8, &dW));
    TSR_TRY(tsr_scratch(ctx, SLOT_OUT, (size_t)(N * T) * 8, &dT));
    TSR_HIP(hipMemcpyAsync(dF, F, (size_t)(N * T * D), hipMemcpyHostToDevice, ctx->stream));
    TSR_HIP(hipMemcpyAsync(dW, LF, (size_t)(L * D) * 8, hipMemcpyHostToDevice, ctx->stream));
    TSR_TRY(tsr_launch_max_detect_examples(ctx, (const uint8_t *)dF, N, T, D, (const double *)dW, L, nullptr, (double *)dT, nullptr));
    std::vector<double> tmp((size_t)(N * T));
    TSR_HIP(hipMemcpyAsync(tmp.data(), dT, tmp.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    TSR_HIP(hipStreamSynchronize(ctx->stream));
    for (int64_t n = 0; n < N; ++n) {  // :69-70: first t initialises, then strict '>'
        double best = 0.0;
        for (int64_t t = 0; t < lengths[n]; ++t) {
            const double v = tmp[(size_t)(n * T + t)];
            if (t == 0 || v > best) best = v;
        }
        out[n] = best;
    }
    return TSR_OK;
}

// ---- front-end --------------------------------------------------------------------
TSR_API int tsr_parts_create(tsr_ctx *ctx, const double *lp, const double *lip, int64_t K,
                             int64_t P0, int64_t P1, int64_t E, tsr_parts **out)
{
    TSR_REQUIRE(ctx && lp && lip && out, "tsr_parts_create: null pointer");
    *out = nullptr;
    TSR_REQUIRE(K >= 1 && P0 >= 1 && P1 >= 1 && E >= 1 && K <= 65535 && P0 * P1 * E <= (1 << 20), "tsr_parts_create: bad geometry");
    TSR_HIP(hipSetDevice(ctx->device));
    const int64_t Q = P0 * P1 * E;
    std::vector<double> tab((size_t)(Q * K * 2));
    for (int64_t k = 0; k < K; ++k)
        for (int64_t q = 0; q < Q; ++q) {
            tab[(size_t)((q * K + k) * 2 + 0)] = lip[k * Q + q];
            tab[(size_t)((q * K + k) * 2 + 1)] = lp[k * Q + q];
        }
    tsr_parts *p = new (std::nothrow) tsr_parts;
    TSR_REQUIRE(p, "tsr_parts_create: out of host memory");
    p->ctx = ctx;
    p->device = ctx->device;
    p->K = K;
    p->P0 = P0;
    p->P1 = P1;
    p->E = E;
    int s = dev_upload(ctx, &p->d_tab, tab.data(), tab.size());
    if (s == TSR_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) s = TSR_ERR_HIP;
    if (s == TSR_OK) s = tsr_parts_mfma_prepare(ctx, p, lp, lip);
    if (s != TSR_OK) {
        tsr_parts_destroy(p);
        return s;
    }
    *out = p;
    return TSR_OK;
}

TSR_API int tsr_parts_set_mode(tsr_parts *p, int mode)
{
    TSR_REQUIRE(p, "tsr_parts_set_mode: null pointer");
    TSR_REQUIRE(mode == TSR_PARTS_AUTO || mode == TSR_PARTS_VALU || mode == TSR_PARTS_MFMA, "tsr_parts_set_mode: unknown mode %d", mode);
    p->mode = mode;
    return TSR_OK;
}

TSR_API int tsr_parts_info(const tsr_parts *p, int *mfma_available, int64_t *locations, int64_t *rechecked)
{
    TSR_REQUIRE(p, "tsr_parts_info: null pointer");
    if (mfma_available) *mfma_available = p->mfma_ok ? 1 : 0;
    if (locations) *locations = p->last_locations;
    if (rechecked) *rechecked = p->last_rechecked;
    return TSR_OK;
}

TSR_API int tsr_parts_destroy(tsr_parts *p)
{
    if (!p) return TSR_OK;
    // never dereference the ctx here: a child may outlive its context (interpreter
    // shutdown destroys objects in arbitrary order); hipFree needs only the device.
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();
    dev_free(p->d_tab);
    tsr_parts_mfma_free(p);
    delete p;
    return TSR_OK;
}

TSR_API int tsr_code_parts(tsr_ctx *ctx, const uint8_t *X, int64_t X0, int64_t X1, int64_t E,
                           const double *lp, const double *lip, int64_t K, int64_t P0, int64_t P1,
                           int threshold, int fast, double *out)
{
    TSR_REQUIRE(ctx && X && lp && lip && out, "tsr_code_parts: null pointer");
    TSR_REQUIRE(X0 >= P0 && X1 >= P1 && P0 >= 1 && P1 >= 1, "tsr_code_parts: patch %lldx%lld larger than map %lldx%lld", (long long)P0, (long long)P1, (long long)X0, (long long)X1);
    TSR_REQUIRE(X0 * X1 * E < ((int64_t)1 << 31), "tsr_code_parts: map too large");
    TSR_REQUIRE(K >= 1 && E >= 1 && K <= 65535 && P0 * P1 * E <= (1 << 20), "tsr_code_parts: bad geometry");
    // one-shot call: the table goes through the ctx scratch (no object, no allocation per call)
    const int64_t Q = P0 * P1 * E;
    std::vector<double> tab((size_t)(Q * K * 2));
    for (int64_t k = 0; k < K; ++k)
        for (int64_t q = 0; q < Q; ++q) {
            tab[(size_t)((q * K + k) * 2 + 0)] = lip[k * Q + q];
            tab[(size_t)((q * K + k) * 2 + 1)] = lp[k * Q + q];
        }
    TSR_HIP(hipSetDevice(ctx->device));
    const int64_t n_loc = (X0 - P0 + 1) * (X1 - P1 + 1);
    void *dX = nullptr, *dO = nullptr, *dT = nullptr;
    int s = tsr_scratch(ctx, SLOT_IN, (size_t)(X0 * X1 * E), &dX);
    if (s == TSR_OK) s = tsr_scratch(ctx, SLOT_OUT, (size_t)(n_loc * (K + 1)) * 8, &dO);
    if (s == TSR_OK) s = tsr_scratch(ctx, SLOT_W, tab.size() * 8, &dT);
    tsr_parts tmp;
    tmp.ctx = ctx;
    tmp.device = ctx->device;
    tmp.K = K;
    tmp.P0 = P0;
    tmp.P1 = P1;
    tmp.E = E;
    tmp.d_tab = (double *)dT;
    if (s == TSR_OK && (hipMemcpyAsync(dX, X, (size_t)(X0 * X1 * E), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
                        hipMemcpyAsync(dT, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess))
        s = TSR_ERR_HIP;
    if (s == TSR_OK) s = tsr_launch_code_parts(ctx, &tmp, (const uint8_t *)dX, X0, X1, threshold, fast, (double *)dO, nullptr);
    if (s == TSR_OK) s = tsr_download(ctx, out, dO, (size_t)(n_loc * (K + 1)) * 8);
    if (s == TSR_ERR_HIP) tsr_set_error("tsr_code_parts: device copy/launch failed");
    return s;
}

// code_parts_mag_aux (code_parts.pyx:138-242): edges + magnitude + auxiliary channels, one-shot.
TSR_API int tsr_code_parts_mag_aux(tsr_ctx *ctx, const uint8_t *X_edges, const uint8_t *X_mags, const uint8_t *X_aux,
                                   int64_t X0, int64_t X1, int64_t E, int64_t MD, int64_t AD, const double *lp,
                                   const double *lip, const double *lmp, const double *lmip, const double *lap,
                                   const double *laip, int64_t K, int64_t P0, int64_t P1, const uint8_t *mask, double *out)
{
    TSR_REQUIRE(ctx && X_edges && X_mags && X_aux && lp && lip && lmp && lmip && lap && laip && mask && out,
                "tsr_code_parts_mag_aux: null pointer");
    TSR_REQUIRE(X0 >= P0 && X1 >= P1 && P0 >= 1 && P1 >= 1 && E >= 1 && K >= 1 && MD >= 1 && AD >= 0,
                "tsr_code_parts_mag_aux: bad geometry (map %lldx%lld, patch %lldx%lld)", (long long)X0, (long long)X1, (long long)P0, (long long)P1);
    TSR_REQUIRE(X0 * X1 * E < ((int64_t)1 << 31) && K <= 65535, "tsr_code_parts_mag_aux: map too large");
    // the six tables as one sequence of {inv, part} pairs in the reference's order of addition
    const int64_t n_terms = P0 * (AD + P1 * E + MD + 1);
    std::vector<double> tab((size_t)(n_terms * K * 2));
    int64_t term = 0;
    auto put = [&](const double *part, const double *inv, int64_t stride, int64_t idx) {
        for (int64_t k = 0; k < K; ++k) {
            tab[(size_t)((term * K + k) * 2 + 0)] = inv[k * stride + idx];
            tab[(size_t)((term * K + k) * 2 + 1)] = part[k * stride + idx];
        }
        ++term;
    };
    for (int64_t i = 0; i < P0; ++i) {
        for (int64_t j = 0; j < AD; ++j) put(lap, laip, P0 * AD, i * AD + j);
        for (int64_t j = 0; j < P1; ++j) {
            for (int64_t z = 0; z < E; ++z) put(lp, lip, P0 * P1 * E, (i * P1 + j) * E + z);
            if (j == P1 / 2)
                for (int64_t mi = 0; mi < MD; ++mi) put(lmp, lmip, P0 + MD, mi + P0);
        }
        put(lmp, lmip, P0 + MD, i);
    }
    TSR_HIP(hipSetDevice(ctx->device));
    const int64_t n_loc = (X0 - P0 + 1) * (X1 - P1 + 1);
    const size_t be = (size_t)(X0 * X1 * E), bm = (size_t)(X0 * MD), ba = (size_t)(X0 * AD), bk = (size_t)(X0 * X1);
    void *dX = nullptr, *dO = nullptr, *dT = nullptr;
    TSR_TRY(tsr_scratch(ctx, SLOT_IN, be + bm + ba + bk + 64, &dX));
    TSR_TRY(tsr_scratch(ctx, SLOT_OUT, (size_t)(n_loc * (K + 1)) * 8, &dO));
    TSR_TRY(tsr_scratch(ctx, SLOT_W, tab.size() * 8, &dT));
    uint8_t *dXe = (uint8_t *)dX, *dXm = dXe + be, *dXa = dXm + bm, *dMk = dXa + ba;
    TSR_HIP(hipMemcpyAsync(dXe, X_edges, be, hipMemcpyHostToDevice, ctx->stream));
    TSR_HIP(hipMemcpyAsync(dXm, X_mags, bm, hipMemcpyHostToDevice, ctx->stream));
    if (ba) TSR_HIP(hipMemcpyAsync(dXa, X_aux, ba, hipMemcpyHostToDevice, ctx->stream));
    TSR_HIP(hipMemcpyAsync(dMk, mask, bk, hipMemcpyHostToDevice, ctx->stream));
    TSR_HIP(hipMemcpyAsync(dT, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    TSR_TRY(tsr_launch_code_parts_mag_aux(ctx, dXe, dXm, dXa, X0, X1, E, MD, AD, P0, P1, dMk, (const double *)dT, K, (double *)dO));
    return tsr_download(ctx, out, dO, (size_t)(n_loc * (K + 1)) * 8);
}

static int upload_part_ids(tsr_ctx *ctx, const int64_t *X, int64_t n, int64_t K, void **dA)
{
    std::vector<int32_t> a((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        TSR_REQUIRE(X[i] <= K, "spread_patches: part id %lld exceeds num_parts=%lld (the reference writes out of bounds here)", (long long)X[i], (long long)K);
        a[(size_t)i] = X[i] > 0 ? (int32_t)X[i] : 0;
    }
    TSR_TRY(tsr_scratch(ctx, SLOT_AUX, (size_t)n * 4, dA));
    TSR_HIP(hipMemcpyAsync(*dA, a.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    TSR_HIP(hipStreamSynchronize(ctx->stream));
    return TSR_OK;
}

TSR_API int tsr_spread_patches(tsr_ctx *ctx, const int64_t *X, int64_t X0, int64_t X1, int r0,
                               int r1, int64_t K, uint8_t *out)
{
    TSR_REQUIRE(ctx && X && out, "tsr_spread_patches: null pointer");
    TSR_REQUIRE(X0 >= 0 && X1 >= 0 && K >= 0 && X0 <= 65535 && X1 <= 65535, "tsr_spread_patches: bad sizes (the reference stores the map shape in uint16)");
    TSR_REQUIRE(r0 >= 0 && r1 >= 0, "tsr_spread_patches: negative spread radius");
    const int64_t n = X0 * X1;
    if (n == 0 || K == 0) return TSR_OK;
    TSR_HIP(hipSetDevice(ctx->device));
    void *dA, *dO;
    TSR_TRY(upload_part_ids(ctx, X, n, K, &dA));
    TSR_TRY(tsr_scratch(ctx, SLOT_OUT, (size_t)(n * K), &dO));
    TSR_TRY(tsr_launch_spread(ctx, (const int32_t *)dA, X0, X1, r0, r1, K, 1, 1, nullptr, (uint8_t *)dO));
    TSR_HIP(hipMemcpyAsync(out, dO, (size_t)(n * K), hipMemcpyDeviceToHost, ctx->stream));
    TSR_HIP(hipStreamSynchronize(ctx->stream));
    return TSR_OK;
}

TSR_API int tsr_spread_graph_patches(tsr_ctx *ctx, uint8_t *X, int64_t X0, int64_t X1, int64_t K,
                                     const uint8_t *graph)
{
    TSR_REQUIRE(ctx && X && graph, "tsr_spread_graph_patches: null pointer");
    const int64_t n = X0 * X1;
    if (n <= 0 || K <= 0) return TSR_OK;
    TSR_HIP(hipSetDevice(ctx->device));
    void *dX, *dG;
    TSR_TRY(tsr_scratch(ctx, SLOT_OUT, (size_t)(n * K), &dX));
    TSR_TRY(tsr_scratch(ctx, SLOT_AUX2, (size_t)(K * K), &dG));
    TSR_HIP(hipMemcpyAsync(dX, X, (size_t)(n * K), hipMemcpyHostToDevice, ctx->stream));
    TSR_HIP(hipMemcpyAsync(dG, graph, (size_t)(K * K), hipMemcpyHostToDevice, ctx->stream));
    TSR_TRY(tsr_launch_spread_graph(ctx, (uint8_t *)dX, n, K, (const uint8_t *)dG));
    TSR_HIP(hipMemcpyAsync(X, dX, (size_t)(n * K), hipMemcpyDeviceToHost, ctx->stream));
    TSR_HIP(hipStreamSynchronize(ctx->stream));
    return TSR_OK;
}

TSR_API int tsr_part_features(tsr_ctx *ctx, const tsr_parts *parts, const uint8_t *X, int64_t X0,
                              int64_t X1, int threshold, int fast, int r0, int r1,
                              const uint8_t *graph, int g0, int g1, uint8_t *out,
                              int64_t *argmax_out)
{
    TSR_REQUIRE(ctx && parts && X && out, "tsr_part_features: null pointer");
    TSR_REQUIRE(X0 >= parts->P0 && X1 >= parts->P1, "tsr_part_features: map smaller than the patch");
    TSR_REQUIRE(r0 >= 0 && r1 >= 0 && g0 >= 1 && g1 >= 1, "tsr_part_features: bad radius/grid");
    TSR_REQUIRE(X0 * X1 * parts->E < ((int64_t)1 << 31), "tsr_part_features: map too large");
    TSR_HIP(hipSetDevice(ctx->device));
    const int64_t N0 = X0 - parts->P0 + 1, N1 = X1 - parts->P1 + 1, K = parts->K;
    const int64_t O0 = (N0 + g0 - 1) / g0, O1 = (N1 + g1 - 1) / g1;
    void *dX, *dA, *dO, *dG = nullptr;
    TSR_TRY(tsr_scratch(ctx, SLOT_IN, (size_t)(X0 * X1 * parts->E), &dX));
    TSR_TRY(tsr_scratch(ctx, SLOT_AUX, (size_t)(N0 * N1) * 4, &dA));
    TSR_TRY(tsr_scratch(ctx, SLOT_OUT, (size_t)(O0 * O1 * K), &dO));
    TSR_HIP(hipMemcpyAsync(dX, X, (size_t)(X0 * X1 * parts->E), hipMemcpyHostToDevice, ctx->stream));
    if (graph) {
        TSR_TRY(tsr_scratch(ctx, SLOT_AUX2, (size_t)(K * K), &dG));
        TSR_HIP(hipMemcpyAsync(dG, graph, (size_t)(K * K), hipMemcpyHostToDevice, ctx->stream));
    }
    TSR_TRY(tsr_launch_code_parts(ctx, parts, (const uint8_t *)dX, X0, X1, threshold, fast, nullptr, (int32_t *)dA));
    TSR_TRY(tsr_launch_spread(ctx, (const int32_t *)dA, N0, N1, r0, r1, K, g0, g1, (const uint8_t *)dG, (uint8_t *)dO));
    TSR_HIP(hipMemcpyAsync(out, dO, (size_t)(O0 * O1 * K), hipMemcpyDeviceToHost, ctx->stream));
    std::vector<int32_t> am;
    if (argmax_out) {
        am.resize((size_t)(N0 * N1));
        TSR_HIP(hipMemcpyAsync(am.data(), dA, am.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    TSR_HIP(hipStreamSynchronize(ctx->stream));
    if (argmax_out)
        for (size_t i = 0; i < am.size(); ++i) argmax_out[i] = am[i];
    return TSR_OK;
}

// Edge-map corpus -> part-map corpus, entirely on the device (BASELINE config 3): for every
// utterance code_parts[_fast] -> argmax -> spread(r0, r1) -> [graph] -> [::g0, ::g1], i.e.
// get_train_data.py:2619-2628 / :183-191 applied to a whole resident frame stream.  The result is
// a new corpus (D = O1*K per frame) that tsr_corpus_score consumes directly.
TSR_API int tsr_corpus_part_features(tsr_ctx *ctx, const tsr_parts *parts, const tsr_corpus *edges,
                                     int threshold, int fast, int r0, int r1, const uint8_t *graph,
                                     int g0, int g1, tsr_corpus **out)
{
    TSR_REQUIRE(ctx && parts && edges && out, "tsr_corpus_part_features: null pointer");
    *out = nullptr;
    TSR_REQUIRE(r0 >= 0 && r1 >= 0 && g0 >= 1 && g1 >= 1, "tsr_corpus_part_features: bad radius/grid");
    TSR_REQUIRE(edges->D % parts->E == 0, "tsr_corpus_part_features: corpus D=%lld is not a multiple of E=%lld", (long long)edges->D, (long long)parts->E);
    const int64_t X1 = edges->D / parts->E, K = parts->K;
    TSR_REQUIRE(X1 >= parts->P1, "tsr_corpus_part_features: %lld frequency bins < patch width %lld", (long long)X1, (long long)parts->P1);
    TSR_HIP(hipSetDevice(ctx->device));
    const int64_t N1 = X1 - parts->P1 + 1, O1 = (N1 + g1 - 1) / g1;
    const int64_t N = edges->N;
    // per-utterance geometry of the output stream
    std::vector<int64_t> ooff((size_t)N + 1, 0);
    for (int64_t n = 0; n < N; ++n) {
        const int64_t T = edges->offsets[n + 1] - edges->offsets[n];
        const int64_t N0 = T >= parts->P0 ? T - parts->P0 + 1 : 0;
        ooff[n + 1] = ooff[n] + (N0 + g0 - 1) / g0;
    }
    const int64_t P_out = ooff[N];
    std::vector<int32_t> ubeg((size_t)std::max<int64_t>(N, 1)), un0((size_t)std::max<int64_t>(N, 1));
    for (int64_t n = 0; n < N; ++n) {
        const int64_t T = edges->offsets[n + 1] - edges->offsets[n];
        ubeg[(size_t)n] = (int32_t)edges->offsets[n];
        un0[(size_t)n] = (int32_t)(T >= parts->P0 ? T - parts->P0 + 1 : 0);
    }
    // the output corpus object (features are produced on the device, not uploaded)
    // Resident form of the result: bit rows (an eighth of the bytes to write here and to read in the bank pass) whenever
    // the dilation kernel can write them; byte rows are made from them on first use by anything that wants bytes.
    const int Dp_new = (int)((O1 * K + 15) / 16 * 16);
    int pw_new = 0, nph_new = 0;
    tsr_bit_geometry(O1 * K, &pw_new, &nph_new);
    const bool want_bits = ctx->opt_part_layout != TSR_PART_LAYOUT_BYTES && tsr_spread_packs_bits(N1, r0, K, Dp_new, pw_new * nph_new);
    TSR_REQUIRE(want_bits || ctx->opt_part_layout != TSR_PART_LAYOUT_BITS,
                "tsr_corpus_part_features: TSR_PART_LAYOUT_BITS, but the dilation cannot write bit rows for K=%lld, r0=%d", (long long)K, r0);
    TSR_TRY(tsr_corpus_need_bytes(ctx, edges));
    // per-utterance tables: ONE upload (output offsets | first source frame | patch rows), everything per frame is made on
    // the device; the host vectors outlive the copies (this call synchronises before it returns)
    void *dGeo = nullptr;
    const size_t geo_bytes = (size_t)(N + 1) * 8 + (size_t)std::max<int64_t>(N, 1) * 8;
    TSR_TRY(tsr_scratch(ctx, SLOT_IN, geo_bytes, &dGeo));
    std::vector<int64_t> geo(geo_bytes / 8, 0);
    memcpy(geo.data(), ooff.data(), (size_t)(N + 1) * 8);
    memcpy(reinterpret_cast<char *>(geo.data()) + (size_t)(N + 1) * 8, ubeg.data(), (size_t)N * 4);
    memcpy(reinterpret_cast<char *>(geo.data()) + (size_t)(N + 1) * 8 + (size_t)N * 4, un0.data(), (size_t)N * 4);
    TSR_HIP(hipMemcpyAsync(dGeo, geo.data(), geo_bytes, hipMemcpyHostToDevice, ctx->stream));
    int64_t *d_ooff = (int64_t *)dGeo;
    int32_t *d_beg = (int32_t *)(d_ooff + N + 1), *d_n0 = d_beg + N;
    tsr_corpus *cp = nullptr;
    {
        const int sc = tsr_corpus_create_empty(ctx, ooff.data(), N, O1 * K, &cp, !want_bits, d_ooff, false);  // (tables: the setup kernel below)
        if (sc != TSR_OK) {
            (void)hipStreamSynchronize(ctx->stream);  // (the upload reads host vectors of this frame)
            return sc;
        }
    }
    if (want_bits) {
        const int sb = tsr_corpus_alloc_bits(ctx, cp, false);
        if (sb != TSR_OK) {
            (void)hipStreamSynchronize(ctx->stream);
            tsr_corpus_destroy(cp);
            return sb;
        }
    }
    void *dA = nullptr, *dRow = nullptr, *dG = nullptr, *dGm = nullptr;
    int s = tsr_scratch(ctx, SLOT_AUX, (size_t)std::max<int64_t>(edges->P * N1, 1) * 4, &dA);
    if (s == TSR_OK) s = tsr_scratch(ctx, SLOT_W, (size_t)std::max<int64_t>(P_out, 1) * 12, &dRow);
    int32_t *d_src_row = (int32_t *)dRow, *d_src_lo = d_src_row + P_out, *d_src_hi = d_src_lo + P_out;
    // the coder's work counters live in their scratch slot from here on (zeroed by the setup kernel)
    const bool try_mfma = edges->P > 0 && parts->mode != TSR_PARTS_VALU && parts->mfma_ok;
    void *dWork = nullptr;
    if (s == TSR_OK && try_mfma) s = tsr_scratch(ctx, SLOT_CNT, 64, &dWork);
    if (s == TSR_OK)  // ONE launch: source geometry, utterance tables of the new corpus, plane tails, work counters
        s = tsr_launch_part_chain_setup(ctx, d_ooff, d_beg, d_n0, N, g0, P_out, d_src_row, d_src_lo, d_src_hi, cp->d_utt_beg, cp->d_utt_end,
                                        reinterpret_cast<uint32_t *>(cp->d_bits), cp->d_bits ? (int64_t)cp->bit_rows * cp->bit_pw : 0,
                                        (int64_t)cp->P * cp->bit_pw, 48 * cp->bit_pw, cp->d_bits ? cp->bit_nph : 0, (int *)dWork);
    std::vector<uint64_t> gmasks;  // graph rows as part masks (K <= 64), behind the K*K bytes
    if (s == TSR_OK && graph) {
        gmasks.assign((size_t)K, 0);
        if (K <= 64)
            for (int64_t a = 0; a < K; ++a)
                for (int64_t b = 0; b < K; ++b)
                    if (graph[a * K + b]) gmasks[(size_t)a] |= 1ull << b;
        const size_t goff = (size_t)((K * K + 15) / 16 * 16);
        s = tsr_scratch(ctx, SLOT_AUX2, goff + (size_t)K * 8, &dG);
        if (s == TSR_OK && (hipMemcpyAsync(dG, graph, (size_t)(K * K), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
                            hipMemcpyAsync((uint8_t *)dG + goff, gmasks.data(), (size_t)K * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess))
            s = TSR_ERR_HIP;
        dGm = (uint8_t *)dG + goff;
    }
    // part ids: the MFMA coder when the tables fit it, else (or on request) the VALU kernel.  In the chain MFMA coder ->
    // mask dilation the id map is BYTES (K <= 63; 8 instead of 33 MB at config 3, written once and read once)
    const int ids8 = tsr_spread_packs_bits(N1, r0, K, Dp_new, want_bits ? pw_new * nph_new : 0) ? 1 : 0;
    bool used_mfma = false;
    if (s == TSR_OK && edges->P > 0) {
        int sm = TSR_ERR_UNSUPPORTED;
        if (parts->mode != TSR_PARTS_VALU && parts->mfma_ok) {
            void *dList = nullptr;
            sm = tsr_scratch(ctx, SLOT_LIST, (size_t)std::max<int64_t>(edges->P * N1, 1) * 8, &dList);
            if (sm == TSR_OK)
                sm = tsr_launch_code_parts_mfma(ctx, parts, edges->d_feats, edges->P, X1, edges->Dp, edges->d_utt_end,
                                                threshold, fast, (int32_t *)dA, (uint32_t *)dList, (int *)dWork, ids8);
            used_mfma = sm == TSR_OK;
        }
        if (sm == TSR_ERR_UNSUPPORTED && parts->mode == TSR_PARTS_MFMA) {
            tsr_set_error("tsr_corpus_part_features: the MFMA coder does not cover this case (%s)",
                          parts->mfma_ok ? "corpus geometry" : parts->mfma_why);
            s = TSR_ERR_UNSUPPORTED;
        } else if (sm == TSR_ERR_UNSUPPORTED) {
            s = tsr_launch_code_parts_stream(ctx, parts, edges->d_feats, edges->P, X1, edges->Dp, edges->d_utt_end,
                                             threshold, fast, (int32_t *)dA);
        } else {
            s = sm;
        }
    }
    if (s == TSR_OK && P_out > 0)
        s = tsr_launch_spread_stream(ctx, (const int32_t *)dA, N1, r0, r1, K, g0, g1, O1, P_out, d_src_row,
                                     d_src_lo, d_src_hi, (const uint8_t *)dG, (const uint64_t *)dGm, cp->Dp, cp->d_feats, cp->d_bits, cp->bit_pw,
                                     cp->bit_nph, cp->bit_rows, used_mfma ? ids8 : 0);
    int work_host[2] = {0, -1};
    if (s == TSR_OK && used_mfma &&
        hipMemcpyAsync(work_host, dWork, sizeof(work_host), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        s = TSR_ERR_HIP;
    if (s == TSR_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) s = TSR_ERR_HIP;  // host staging dies here
    if (s == TSR_OK) {  // bookkeeping for tsr_parts_info (the handle is const for the caller's sake only)
        tsr_parts *pm = const_cast<tsr_parts *>(parts);
        pm->last_locations = edges->P * N1;
        pm->last_rechecked = used_mfma ? work_host[1] : -1;
    }
    if (s != TSR_OK) {
        if (s == TSR_ERR_HIP) tsr_set_error("tsr_corpus_part_features: device copy/launch failed");
        (void)hipStreamSynchronize(ctx->stream);  // (nothing of this frame may still be read by a copy)
        tsr_corpus_destroy(cp);
        return s;
    }
    *out = cp;
    return TSR_OK;
}

TSR_API int tsr_corpus_download_feats(tsr_ctx *ctx, const tsr_corpus *cp, uint8_t *feats)
{
    TSR_REQUIRE(ctx && cp && (feats || cp->P == 0), "tsr_corpus_download_feats: null pointer");
    TSR_HIP(hipSetDevice(ctx->device));
    TSR_TRY(tsr_corpus_need_bytes(ctx, cp));
    if (cp->P)
        TSR_HIP(hipMemcpy2DAsync(feats, (size_t)cp->D, cp->d_feats, (size_t)cp->Dp, (size_t)cp->D, (size_t)cp->P,
                                 hipMemcpyDeviceToHost, ctx->stream));
    TSR_HIP(hipStreamSynchronize(ctx->stream));
    return TSR_OK;
}

TSR_API int tsr_corpus_layout(const tsr_corpus *cp, int *layout, int64_t *feature_bytes)
{
    TSR_REQUIRE(cp, "tsr_corpus_layout: null corpus");
    if (layout) *layout = (cp->d_feats ? 1 : 0) | (cp->d_bits ? 2 : 0);
    if (feature_bytes) *feature_bytes = (int64_t)((cp->d_feats ? cp->feats_alloc : 0) + (cp->d_bits ? cp->bits_alloc : 0));
    return TSR_OK;
}

TSR_API int tsr_corpus_shape(const tsr_corpus *cp, int64_t *N, int64_t *D, int64_t *frame_offsets)
{
    TSR_REQUIRE(cp, "tsr_corpus_shape: null corpus");
    if (N) *N = cp->N;
    if (D) *D = cp->D;
    if (frame_offsets) memcpy(frame_offsets, cp->offsets.data(), (size_t)(cp->N + 1) * sizeof(int64_t));
    return TSR_OK;
}

// ---- multi-GPU: RCCL bank broadcast (librccl resolved at run time) -------------------
namespace {
struct uid128 {  // ncclUniqueId: 128 opaque bytes, passed by value
    char b[128];
};
struct rccl_api {
    void *h = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, uid128, int) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
rccl_api g_rccl;
}  // namespace

static int rccl_load()
{
    if (g_rccl.h) return TSR_OK;
    // Take the librccl that sits next to the HIP runtime this process actually uses: a process
    // may hold two ROCm stacks (e.g. /opt/rocm and the one bundled with a torch wheel, same
    // sonames), and an RCCL built against the other runtime fails in ncclCommInitRank.
    std::string dir;
    Dl_info info;
    if (dladdr(reinterpret_cast<void *>(&hipGetDeviceCount), &info) && info.dli_fname) {
        dir = info.dli_fname;
        const size_t slash = dir.rfind('/');
        dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);
    }
    std::vector<std::string> names;
    if (!dir.empty()) {
        names.push_back(dir + "librccl.so.1");
        names.push_back(dir + "librccl.so");
    }
    names.push_back("librccl.so.1");
    names.push_back("librccl.so");
    void *h = nullptr;
    for (const std::string &n : names)
        if ((h = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) {
        tsr_set_error("RCCL: cannot dlopen librccl (%s)", dlerror());
        return TSR_ERR_RCCL;
    }
    g_rccl.GetUniqueId = (int (*)(void *))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(void **, int, uid128, int))dlsym(h, "ncclCommInitRank");
    g_rccl.Broadcast = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(h, "ncclBroadcast");
    g_rccl.AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(h, "ncclAllReduce");
    g_rccl.CommDestroy = (int (*)(void *))dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.Broadcast || !g_rccl.AllReduce || !g_rccl.CommDestroy) {
        tsr_set_error("RCCL: librccl lacks a required symbol");
        dlclose(h);
        return TSR_ERR_RCCL;
    }
    g_rccl.h = h;
    return TSR_OK;
}

#define TSR_NCCL(call)                                                                      \
    do {                                                                                    \
        int _r = (call);                                                                    \
        if (_r != 0) {                                                                      \
            tsr_set_error("%s failed: %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "?"); \
            return TSR_ERR_RCCL;                                                            \
        }                                                                                   \
    } while (0)

TSR_API int tsr_dist_unique_id(char id[128])
{
    TSR_REQUIRE(id, "tsr_dist_unique_id: null pointer");
    TSR_TRY(rccl_load());
    TSR_NCCL(g_rccl.GetUniqueId(id));
    return TSR_OK;
}

TSR_API int tsr_dist_init(tsr_ctx *ctx, const char id[128], int nranks, int rank)
{
    TSR_REQUIRE(ctx && id, "tsr_dist_init: null pointer");
    TSR_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "tsr_dist_init: bad rank %d of %d", rank, nranks);
    TSR_REQUIRE(!ctx->comm, "tsr_dist_init: communicator already initialised");
    TSR_TRY(rccl_load());
    TSR_HIP(hipSetDevice(ctx->device));
    uid128 u;
    memcpy(u.b, id, 128);
    if (!ctx->d_coll) TSR_HIP(hipMalloc(&ctx->d_coll, 64));  // header + verdict flag of tsr_bank_broadcast
    TSR_NCCL(g_rccl.CommInitRank(&ctx->comm, nranks, u, rank));
    ctx->nranks = nranks;
    ctx->rank = rank;
    return TSR_OK;
}

TSR_API int tsr_dist_finalize(tsr_ctx *ctx)
{
    TSR_REQUIRE(ctx, "tsr_dist_finalize: null ctx");
    if (ctx->comm && g_rccl.CommDestroy) {
        (void)hipStreamSynchronize(ctx->stream);
        g_rccl.CommDestroy(ctx->comm);
    }
    ctx->comm = nullptr;
    if (ctx->d_coll) (void)hipFree(ctx->d_coll);
    ctx->d_coll = nullptr;
    ctx->nranks = 1;
    ctx->rank = 0;
    return TSR_OK;
}

TSR_API int tsr_bank_broadcast(tsr_ctx *ctx, tsr_bank *bank, int root)
{
    TSR_REQUIRE(ctx && bank, "tsr_bank_broadcast: null pointer");
    TSR_REQUIRE(ctx->comm, "tsr_bank_broadcast: call tsr_dist_init first");
    TSR_REQUIRE(root >= 0 && root < ctx->nranks, "tsr_bank_broadcast: bad root %d", root);
    TSR_HIP(hipSetDevice(ctx->device));
    // Canonical form travels: [L i32 x M | pad][c f64 x M][W_cat f32 x sumL*D]; every rank
    // then rebuilds its kernel-side layouts from it.
    const size_t nW = bank->W_cat.size();
    const size_t offC = ((size_t)bank->M * 4 + 7) & ~(size_t)7;
    const size_t offW = offC + (size_t)bank->M * 8;
    const size_t bytes = offW + nW * 4;
    // 1) geometry header {M, D, sumL, hash of L[]} from the root; every rank compares it with its own
    //    bank and the verdicts are MIN-reduced, so that either all ranks go on to the payload (whose
    //    byte count then agrees everywhere) or all of them return TSR_ERR_INVALID -- a mismatch can
    //    neither hang the collective nor corrupt a bank.
    int64_t hdr[4] = {bank->M, bank->D, bank->sumL, 1469598103934665603ll};
    for (int32_t l : bank->L) hdr[3] = (hdr[3] ^ (int64_t)l) * 1099511628211ll;  // FNV-1a over the lengths
    // Every rank walks through the SAME collectives -- header broadcast, verdict all-reduce, [payload broadcast, final
    // verdict all-reduce] -- whatever fails locally in between: a local failure (allocation, copy, rebuild) is carried
    // into the next verdict as 0 instead of returning early, so that no rank is left waiting inside a collective the
    // failing rank never enters and all ranks return an error together.  Only a failing RCCL call itself returns at
    // once (the communicator is then unusable anyway).
    int local = TSR_OK;  // first local failure
    auto note = [&](int st) { if (local == TSR_OK && st != TSR_OK) local = st; };
    auto hipst = [&](hipError_t e) { return e == hipSuccess ? TSR_OK : TSR_ERR_HIP; };
    void *d = nullptr;
    note(tsr_scratch(ctx, SLOT_W, std::max<size_t>(bytes, 64), &d));
    char *dc = d ? static_cast<char *>(d) : static_cast<char *>(ctx->d_coll);  // (the 64-byte buffer of tsr_dist_init)
    int32_t *d_flag = reinterpret_cast<int32_t *>(dc + 32);
    auto verdict = [&](int32_t mine, int32_t *all) -> int {  // MIN over ranks of `mine`
        *all = 0;
        note(hipst(hipMemcpyAsync(d_flag, &mine, 4, hipMemcpyHostToDevice, ctx->stream)));
        TSR_NCCL(g_rccl.AllReduce(d_flag, d_flag, 1, /*ncclInt32*/ 2, /*ncclMin*/ 3, ctx->comm, ctx->stream));
        int32_t v = 0;
        note(hipst(hipMemcpyAsync(&v, d_flag, 4, hipMemcpyDeviceToHost, ctx->stream)));
        note(hipst(hipStreamSynchronize(ctx->stream)));
        *all = (local == TSR_OK) ? v : 0;
        return TSR_OK;
    };
    note(hipst(hipMemcpyAsync(dc, hdr, sizeof hdr, hipMemcpyHostToDevice, ctx->stream)));
    TSR_NCCL(g_rccl.Broadcast(dc, dc, sizeof hdr, /*ncclChar*/ 0, root, ctx->comm, ctx->stream));
    int64_t got[4] = {0, 0, 0, 0};
    note(hipst(hipMemcpyAsync(got, dc, sizeof got, hipMemcpyDeviceToHost, ctx->stream)));
    note(hipst(hipStreamSynchronize(ctx->stream)));
    const int32_t same = (local == TSR_OK && d && memcmp(got, hdr, sizeof hdr) == 0) ? 1 : 0;
    int32_t all_same = 0;
    TSR_TRY(verdict(same, &all_same));
    if (!all_same) {
        if (local != TSR_OK || !d) {
            tsr_set_error("tsr_bank_broadcast: rank %d could not stage the bank (device allocation or copy failed)", ctx->rank);
            return local != TSR_OK ? local : TSR_ERR_HIP;
        }
        if (same)
            tsr_set_error("tsr_bank_broadcast: another rank's bank geometry differs from root %d's (or it failed locally)", root);
        else
            tsr_set_error("tsr_bank_broadcast: rank %d's bank geometry (M=%d D=%lld sumL=%lld, template lengths) "
                          "differs from root's (M=%lld D=%lld sumL=%lld)", ctx->rank, bank->M, (long long)bank->D,
                          (long long)bank->sumL, (long long)got[0], (long long)got[1], (long long)got[2]);
        return TSR_ERR_INVALID;
    }
    // 2) the payload
    std::vector<char> host(bytes, 0);
    memcpy(host.data(), bank->L.data(), (size_t)bank->M * 4);
    memcpy(host.data() + offC, bank->c.data(), (size_t)bank->M * 8);
    memcpy(host.data() + offW, bank->W_cat.data(), nW * 4);
    note(hipst(hipMemcpyAsync(d, host.data(), bytes, hipMemcpyHostToDevice, ctx->stream)));
    TSR_NCCL(g_rccl.Broadcast(d, d, bytes, /*ncclChar*/ 0, root, ctx->comm, ctx->stream));
    std::vector<char> back(bytes);
    note(hipst(hipMemcpyAsync(back.data(), d, bytes, hipMemcpyDeviceToHost, ctx->stream)));
    note(hipst(hipStreamSynchronize(ctx->stream)));
    bool lengths_ok = local == TSR_OK && memcmp(back.data(), bank->L.data(), (size_t)bank->M * 4) == 0;
    if (lengths_ok && ctx->rank != root) {
        memcpy(bank->c.data(), back.data() + offC, (size_t)bank->M * 8);
        memcpy(bank->W_cat.data(), back.data() + offW, nW * 4);
        tsr_fast_bank_free(bank);  // limb tables of the old weights go first: rebuilt lazily from the new ones
        note(bank_build_exact(ctx, bank));
    }
    // 3) final verdict: every rank holds the root's bank, or all of them say so
    int32_t all_done = 0;
    TSR_TRY(verdict(lengths_ok && local == TSR_OK ? 1 : 0, &all_done));
    if (!all_done) {
        if (local != TSR_OK) return local;  // (its own message stands)
        if (!lengths_ok)
            tsr_set_error("tsr_bank_broadcast: rank %d's template lengths differ from root's", ctx->rank);
        else
            tsr_set_error("tsr_bank_broadcast: another rank failed to take over the bank");
        return TSR_ERR_INVALID;
    }
    return TSR_OK;
}
