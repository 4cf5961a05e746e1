// tip_core.hip -- context, memory, error and profiling plumbing of libtissue_hip.so
#include "tip_internal.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>

namespace tip {

static thread_local Ctx *g_ctx = nullptr;
static thread_local int g_requested_device = 0;

static int ctx_create(int device)
{
    Ctx *c = new Ctx();
    c->device = device;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) {
        c->err = std::string("hipSetDevice failed: ") + hipGetErrorString(e);
        g_ctx = c;
        return TIP_ERR_HIP;
    }
    {
        // TIP_STREAM_PRIORITY=low | high: the library's streams below / above the default priority (an experiment hook: with the
        // U-Net's convolutions on torch's default-priority streams, "low" lets the short projection / tail kernels fill gaps only)
        const char *pr = getenv("TIP_STREAM_PRIORITY");
        int lo = 0, hi = 0;
        if (pr && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && (strcmp(pr, "low") == 0 || strcmp(pr, "high") == 0))
            e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, strcmp(pr, "low") == 0 ? lo : hi);
        else
            e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    }
    if (e != hipSuccess) {
        c->err = std::string("hipStreamCreate failed: ") + hipGetErrorString(e);
        c->stream = nullptr;
        g_ctx = c;
        return TIP_ERR_HIP;
    }
    g_ctx = c;
    return TIP_OK;
}

Ctx &ctx()
{
    if (!g_ctx) ctx_create(g_requested_device);
    return *g_ctx;
}

int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    ctx().err = buf;
    return code;
}

void *ws_alloc(size_t bytes)
{
    Ctx &c = ctx();
    bytes = (bytes + 255) & ~(size_t)255;
    int best = -1;
    for (size_t i = 0; i < c.pool.size(); i++)
        if (!c.pool[i].used && c.pool[i].bytes >= bytes && (best < 0 || c.pool[i].bytes < c.pool[best].bytes))
            best = (int)i;
    if (best >= 0 && c.pool[best].bytes <= 2 * bytes + (1 << 20)) {
        c.pool[best].used = true;
        return c.pool[best].p;
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        // release cached free blocks and retry once
        for (auto &b : c.pool)
            if (!b.used && b.p) { (void)hipFree(b.p); b.p = nullptr; b.bytes = 0; }
        e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            fail(TIP_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
            return nullptr;
        }
    }
    c.pool.push_back({p, bytes, true});
    return p;
}

void *pinned_scratch(size_t bytes)
{
    Ctx &c = ctx();
    if (c.pin_bytes >= bytes && c.pin_buf) return c.pin_buf;
    if (c.pin_buf) { (void)hipHostFree(c.pin_buf); c.pin_buf = nullptr; c.pin_bytes = 0; }
    const size_t want = bytes + (bytes >> 2) + 4096;          // (headroom: marker counts vary from frame to frame)
    void *p = nullptr;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        fail(TIP_ERR_NOMEM, "hipHostMalloc(%zu) failed", want);
        return nullptr;
    }
    c.pin_buf = p;
    c.pin_bytes = want;
    return p;
}

void ws_free(void *p)
{
    Ctx &c = ctx();
    for (auto &b : c.pool)
        if (b.p == p) { b.used = false; return; }
}

void prof_begin(const char *name)
{
    Ctx &c = ctx();
    if (!c.prof) return;
    ProfRec r;
    r.name = name;
    if (c.free_events.size() >= 2) {
        r.e0 = c.free_events.back(); c.free_events.pop_back();
        r.e1 = c.free_events.back(); c.free_events.pop_back();
    } else {
        (void)hipEventCreate(&r.e0);
        (void)hipEventCreate(&r.e1);
    }
    (void)hipEventRecord(r.e0, c.stream);
    c.recs.push_back(r);
}

void prof_end()
{
    Ctx &c = ctx();
    if (!c.prof) return;
    (void)hipEventRecord(c.recs.back().e1, c.stream);
}

// ---- tuning table ---------------------------------------------------------------------------------------------------------
namespace {
Tuning g_tuning;
std::once_flag g_tuning_once;
std::mutex g_tuning_mu;

// one assignment from its textual form (nullptr / "": back to the default); false: unknown name
bool tuning_assign(Tuning &t, const char *name, const char *value)
{
    const Tuning def;
    const bool unset = !value || !*value;
    auto flag = [&](int &field, int dflt) { field = unset ? dflt : (strcmp(value, "0") != 0 ? 1 : 0); };
    auto num = [&](int &field, int dflt) { field = unset ? dflt : atoi(value); };
    auto pair = [&](int &a, int &b, int da, int db) {
        a = da; b = db;
        if (!unset) sscanf(value, "%d,%d", &a, &b);
    };
    const std::string n(name ? name : "");
    if (n == "TIP_WS_TIES") {
        if (unset) t.ws_ties = def.ws_ties;
        else t.ws_ties = (!strcmp(value, "fast") || !strcmp(value, "0")) ? 0 : 1;
    }
    else if (n == "TIP_WS_TILE") num(t.ws_tile, def.ws_tile);
    else if (n == "TIP_WS_OPEN") pair(t.ws_open_a, t.ws_open_b, def.ws_open_a, def.ws_open_b);
    else if (n == "TIP_WS_CERT_FROM") num(t.ws_cert_from, def.ws_cert_from);
    else if (n == "TIP_WS_NO_SKIP") flag(t.ws_no_skip, 0);
    else if (n == "TIP_WS_LDS_PAD") num(t.ws_lds_pad, 0);
    else if (n == "TIP_WS_DEBUG") flag(t.ws_debug, 0);
    else if (n == "TIP_WS_NO_ENDGAME") flag(t.ws_no_endgame, 0);
    else if (n == "TIP_WS_NO_WIDE") flag(t.ws_no_wide, 0);
    else if (n == "TIP_MFMA_BLOCKS_PER_CU") { num(t.mfma_blocks_per_cu, def.mfma_blocks_per_cu); t.mfma_blocks_per_cu = std::max(1, std::min(2, t.mfma_blocks_per_cu)); }
    else if (n == "TIP_PROJECT_GENERIC") flag(t.project_generic, 0);
    else if (n == "TIP_PROJECT_UNFUSED_PREBLUR") flag(t.project_unfused_preblur, 0);
    else if (n == "TIP_PROJECT_UNFUSED_MASK") flag(t.project_unfused_mask, 0);
    else if (n == "TIP_PROJECT_EXACT_SCORE") flag(t.project_exact_score, 0);
    else if (n == "TIP_PROJECT_DEBUG") flag(t.project_debug, 0);
    else if (n == "TIP_FAST_CFG") pair(t.fast_cfg_y, t.fast_cfg_x, def.fast_cfg_y, def.fast_cfg_x);
    else if (n == "TIP_UNET_TILE8") num(t.unet_tile8, def.unet_tile8);
    else if (n == "TIP_UF_ONE_LEVEL") flag(t.uf_one_level, 0);
    else if (n == "TIP_UNET_SPB") num(t.unet_spb, def.unet_spb);
    else if (n == "TIP_UNET_XCD_MAP") flag(t.unet_xcd_map, 1);
    else if (n == "TIP_UNET_TAIL_UNFUSED") flag(t.unet_tail_unfused, 0);
    else if (n == "TIP_MB_SMALL") num(t.mb_small, def.mb_small);
    else if (n == "TIP_MB_BATCH") num(t.mb_batch, def.mb_batch);
    else return false;
    return true;
}

const char *const TUNING_NAMES[] = {"TIP_WS_TIES", "TIP_WS_TILE", "TIP_WS_OPEN", "TIP_WS_CERT_FROM", "TIP_WS_NO_SKIP", "TIP_WS_LDS_PAD",
                                    "TIP_WS_DEBUG", "TIP_WS_NO_ENDGAME", "TIP_WS_NO_WIDE", "TIP_MFMA_BLOCKS_PER_CU", "TIP_PROJECT_GENERIC",
                                    "TIP_PROJECT_UNFUSED_PREBLUR", "TIP_PROJECT_UNFUSED_MASK", "TIP_PROJECT_EXACT_SCORE", "TIP_PROJECT_DEBUG",
                                    "TIP_FAST_CFG", "TIP_UNET_TILE8", "TIP_UNET_TAIL_UNFUSED", "TIP_UNET_XCD_MAP", "TIP_UNET_SPB", "TIP_UF_ONE_LEVEL", "TIP_MB_SMALL", "TIP_MB_BATCH"};

void tuning_from_env()
{
    for (const char *name : TUNING_NAMES)
        if (const char *e = getenv(name)) tuning_assign(g_tuning, name, e);
}
}  // namespace

const Tuning &tuning()
{
    std::call_once(g_tuning_once, tuning_from_env);
    return g_tuning;
}

int make_taps(Taps &t, const double *w, int n)
{
    if (n <= 0 || n > 255 || !(n & 1)) return fail(TIP_ERR_ARG, "tap count %d must be odd and <= 255", n);
    for (int i = 1; i <= n / 2; i++)
        if (fabs(w[n / 2 + i] - w[n / 2 - i]) > 2.220446049250313e-16)
            return fail(TIP_ERR_UNSUPPORTED, "only symmetric kernels (scipy's symmetric correlate1d branch)");
    memset(&t, 0, sizeof t);
    for (int i = 0; i < n; i++) t.w[i] = w[i];
    t.n = n;
    return TIP_OK;
}

// numpy's pairwise float64 sum, so that phi/phi.sum() (scipy/ndimage/filters.py:_gaussian_kernel1d) matches
static double np_pairwise_sum(const double *a, long n)
{
    if (n < 8) {
        double r = 0.0;
        for (long i = 0; i < n; i++) r += a[i];
        return r;
    } else if (n <= 128) {
        double r[8];
        long i;
        for (i = 0; i < 8; i++) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    long n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

int libm_taps(double sigma, double truncate, double *w, int cap)
{
    long radius = (long)(truncate * sigma + 0.5);
    long n = 2 * radius + 1;
    if (n > cap) return -1;
    double s2 = sigma * sigma;
    for (long i = 0; i < n; i++) {
        double x = (double)(i - radius);
        w[i] = exp(-0.5 / s2 * (x * x));
    }
    double sum = np_pairwise_sum(w, n);
    for (long i = 0; i < n; i++) w[i] = w[i] / sum;
    return (int)n;
}

}  // namespace tip

using namespace tip;

extern "C" {

int tip_version(void) { return 100; }

int tip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int tip_init(int device)
{
    if (g_ctx) {
        if (g_ctx->device == device && g_ctx->stream) return TIP_OK;
        tip_shutdown();
    }
    g_requested_device = device;
    int rc = ctx_create(device);
    return rc;
}

int tip_shutdown(void)
{
    if (!g_ctx) return TIP_OK;
    Ctx *c = g_ctx;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto &b : c->pool)
        if (b.p) (void)hipFree(b.p);
    for (auto &r : c->recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    for (auto &e : c->free_events) (void)hipEventDestroy(e);
    if (c->edge_event) (void)hipEventDestroy(c->edge_event);
    if (c->prep_ws) (void)hipFree(c->prep_ws);
    if (c->pin_buf) (void)hipHostFree(c->pin_buf);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    g_ctx = nullptr;
    return TIP_OK;
}

int tip_last_error(char *buf, size_t n)
{
    if (!buf || !n) return TIP_ERR_ARG;
    const std::string &e = ctx().err;
    snprintf(buf, n, "%s", e.c_str());
    return TIP_OK;
}

int tip_malloc(void **dptr, size_t bytes)
{
    if (!dptr) return fail(TIP_ERR_ARG, "tip_malloc: null out pointer");
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    TIP_HIP(hipMalloc(dptr, bytes ? bytes : 16));
    return TIP_OK;
}

int tip_free(void *dptr)
{
    if (!dptr) return TIP_OK;
    Ctx &c = ctx();
    if (c.stream) TIP_HIP(hipStreamSynchronize(c.stream));
    TIP_HIP(hipFree(dptr));
    return TIP_OK;
}

int tip_memcpy_h2d(void *dst, const void *src, size_t bytes)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    TIP_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c.stream));
    TIP_HIP(hipStreamSynchronize(c.stream));
    return TIP_OK;
}

int tip_memcpy_d2h(void *dst, const void *src, size_t bytes)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    TIP_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c.stream));
    TIP_HIP(hipStreamSynchronize(c.stream));
    return TIP_OK;
}

int tip_memcpy_d2d(void *dst, const void *src, size_t bytes)   // asynchronous on the calling thread's stream
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    TIP_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c.stream));
    return TIP_OK;
}

// 2-D block copy on the device (a window of a frame into a contiguous buffer): pitches and width in bytes
int tip_memcpy2d_d2d(void *dst, size_t dst_pitch, const void *src, size_t src_pitch, size_t width_bytes, size_t height)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    if (!dst || !src || width_bytes > dst_pitch || width_bytes > src_pitch) return fail(TIP_ERR_ARG, "tip_memcpy2d_d2d: bad arguments");
    if (width_bytes == 0 || height == 0) return TIP_OK;
    TIP_HIP(hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, width_bytes, height, hipMemcpyDeviceToDevice, c.stream));
    return TIP_OK;
}

int tip_memset(void *dst, int value, size_t bytes)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    TIP_HIP(hipMemsetAsync(dst, value, bytes, c.stream));
    return TIP_OK;
}

int tip_sync(void)
{
    Ctx &c = ctx();
    if (!c.stream) return TIP_ERR_HIP;
    TIP_HIP(hipStreamSynchronize(c.stream));
    return TIP_OK;
}

int tip_prof_enable(int on)
{
    ctx().prof = on != 0;
    return TIP_OK;
}

int tip_prof_reset(void)
{
    Ctx &c = ctx();
    if (c.stream) (void)hipStreamSynchronize(c.stream);
    for (auto &r : c.recs) { c.free_events.push_back(r.e0); c.free_events.push_back(r.e1); }
    c.recs.clear();
    return TIP_OK;
}

int tip_prof_report(char *buf, size_t n)
{
    Ctx &c = ctx();
    if (c.stream) (void)hipStreamSynchronize(c.stream);
    std::map<std::string, std::pair<long, double>> agg;
    std::vector<std::string> order;
    for (auto &r : c.recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) ms = 0.f;
        auto it = agg.find(r.name);
        if (it == agg.end()) { agg[r.name] = {1, (double)ms}; order.push_back(r.name); }
        else { it->second.first++; it->second.second += ms; }
    }
    std::string s;
    char line[256];
    for (auto &k : order) {
        snprintf(line, sizeof line, "%s %ld %.6f\n", k.c_str(), agg[k].first, agg[k].second);
        s += line;
    }
    if (buf && n) snprintf(buf, n, "%s", s.c_str());
    return (int)s.size() + 1;
}

// name: one of the TIP_* tuning names (include/tissue_hip.h); value: its textual form, NULL or "" for the default.
// The table is read from the environment once, when the library first needs it; afterwards this is the only way in.
// QUIESCENT USE ONLY: the entry points read the table's plain ints without a lock (one aligned int each: a reader sees the old or the
// new value of a hook, never a torn one, but a call that is in flight while a hook changes may run partly under each) -- the hooks are
// test / experiment switches; set them while no other thread is inside the library.
int tip_set_tuning(const char *name, const char *value)
{
    (void)tuning();
    std::lock_guard<std::mutex> lock(g_tuning_mu);
    if (!tuning_assign(g_tuning, name, value)) return fail(TIP_ERR_ARG, "tip_set_tuning: unknown name %s", name ? name : "(null)");
    return TIP_OK;
}

int tip_gaussian_taps(double sigma, double truncate, double *taps, int cap)
{
    if (!taps || sigma <= 0) return fail(TIP_ERR_ARG, "tip_gaussian_taps: bad arguments");
    int n = libm_taps(sigma, truncate, taps, cap);
    if (n < 0) return fail(TIP_ERR_OVERFLOW, "tip_gaussian_taps: capacity %d too small", cap);
    return n;
}

}  // extern "C"
