// C ABI of libpmk_hip.so (declared in include/pmk.h): contexts, models, queries.
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "pmk_internal.h"

namespace pmk {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int bsp_build(int D, int64_t N, const double *X, int levels, int sign_mode, int dot_mode, BspArrays &t);
int bsp_from_hyperplanes(int D, int levels, const double *hp_v, const double *hp_c, int dot_mode, BspArrays &t);
int bsp_build_device(pmk_ctx *c, int D, int64_t N, const double *X, int levels, int sign_mode, int dot_mode, BspArrays &t);
int bsp_assign_device(pmk_ctx *c, const BspArrays &t, int64_t N, const double *X, double eps, int64_t *offsets,
                      int64_t *inds, int64_t *list_offsets, int64_t *lists);
int64_t bsp_find(const BspArrays &t, const double *x);
int bsp_assign(const BspArrays &t, int64_t N, const double *X, double eps, int64_t *offsets, int64_t *inds,
               int64_t *list_offsets, int64_t *lists);
int64_t bsp_neighbours(const BspArrays &t, const double *p, double radius, double delta, int64_t home,
                       int64_t *region_inds, double *ts, double *zs, uint8_t *keep);

static bool kernel_ok(const pmk_kernel_desc *th)
{
    if (!th) return false;
    switch (th->family) {
    case PMK_SPLINE34: case PMK_SPLINE12: case PMK_SPLINE32: case PMK_GAUSSIAN: case PMK_RQ: case PMK_TRQ:
    case PMK_MODSQEXP: case PMK_BB10: case PMK_BB20: case PMK_BB1EPS: case PMK_BB2EPS:
        return true;
    default:
        return false;
    }
}

template <typename T>
static int dev_alloc(T **p, int64_t count)
{
    *p = nullptr;
    if (count <= 0) count = 1;
    PMK_HIP(hipMalloc((void **)p, sizeof(T) * (size_t)count));
    return 0;
}

template <typename T>
static void dev_free(T *&p)
{
    if (p) (void)hipFree((void *)p);
    p = nullptr;
}

// device temporary of a blocking host-API call: released on every return path
template <typename T>
struct DevTmp {
    T *p = nullptr;
    DevTmp() = default;
    DevTmp(const DevTmp &) = delete;
    DevTmp &operator=(const DevTmp &) = delete;
    ~DevTmp() { if (p) (void)hipFree((void *)p); }
    int alloc(int64_t count) { return dev_alloc(&p, count); }
    operator T *() const { return p; }
};

// host double buffer -> device buffer of the model's element type (and back)
static int upload_real(const pmk_model *m, void *dst, int64_t elem_off, const double *src, size_t count)
{
    if (m->dtype == PMK_F32) {
        std::vector<float> tmp(count);
        for (size_t i = 0; i < count; ++i) tmp[i] = (float)src[i];
        PMK_HIP(hipMemcpy((char *)dst + elem_off * 4, tmp.data(), 4 * count, hipMemcpyHostToDevice));
    } else {
        PMK_HIP(hipMemcpy((char *)dst + elem_off * 8, src, 8 * count, hipMemcpyHostToDevice));
    }
    return 0;
}
// rows x cols block with device leading dimension ldd -> host leading dimension ldh (synchronous)
static int download_real_2d(const pmk_model *m, double *dst, int64_t ldh, const void *src, int64_t elem_off, int64_t ldd,
                            int64_t rows, int64_t cols, hipStream_t s)
{
    if (m->dtype == PMK_F32) {
        std::vector<float> tmp((size_t)(rows * cols));
        PMK_HIP(hipMemcpy2DAsync(tmp.data(), 4 * rows, (const char *)src + elem_off * 4, 4 * ldd, 4 * rows, (size_t)cols,
                                 hipMemcpyDeviceToHost, s));
        PMK_HIP(hipStreamSynchronize(s));
        for (int64_t j = 0; j < cols; ++j)
            for (int64_t i = 0; i < rows; ++i) dst[i + j * ldh] = (double)tmp[(size_t)(i + j * rows)];
    } else {
        PMK_HIP(hipMemcpy2DAsync(dst, 8 * ldh, (const char *)src + elem_off * 8, 8 * ldd, 8 * rows, (size_t)cols,
                                 hipMemcpyDeviceToHost, s));
        PMK_HIP(hipStreamSynchronize(s));
    }
    return 0;
}

// point-major host points (D x n) -> SoA rows of length ld.  Padding entries are 1e300: their distance
// to any real point overflows to +inf, so a compactly supported profile evaluates to exactly 0 there.
static void pack_soa(int D, int64_t n, int64_t ld, const double *X, double *out)
{
    for (int d = 0; d < D; ++d) {
        double *row = out + (int64_t)d * ld;
        for (int64_t i = 0; i < n; ++i) row[i] = X[i * D + d];
        for (int64_t i = n; i < ld; ++i) row[i] = 1e300;
    }
}

}  // namespace pmk

using namespace pmk;

void pmk_ctx::tic(const char *name)
{
    if (!timers) return;
    for (auto &t : tm)
        if (t.name == name) { (void)hipEventRecord(t.a, stream); t.valid = false; return; }
    Timer t;
    t.name = name;
    (void)hipEventCreate(&t.a);
    (void)hipEventCreate(&t.b);
    t.valid = false;
    (void)hipEventRecord(t.a, stream);
    tm.push_back(t);
}

void pmk_ctx::toc(const char *name)
{
    if (!timers) return;
    for (auto &t : tm)
        if (t.name == name) { (void)hipEventRecord(t.b, stream); t.valid = true; return; }
}

extern "C" {

int pmk_version(void) { return PMK_VERSION; }
const char *pmk_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------ context
int pmk_ctx_create(int device, pmk_ctx **out)
{
    if (!out) { set_error("pmk_ctx_create: out is NULL"); return -2; }
    *out = nullptr;
    int ndev = 0;
    PMK_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) { set_error("pmk_ctx_create: device %d of %d", device, ndev); return -1; }
    PMK_HIP(hipSetDevice(device));
    pmk_ctx *c = new (std::nothrow) pmk_ctx();
    if (!c) { set_error("out of memory"); return -100; }
    c->device = device;
    hipDeviceProp_t prop;
    // the context's own stream gets the highest priority, the side stream of the pipelined kernel-matrix build the
    // lowest: its workgroups then only take what the factorisation's launches leave free
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    if (hipGetDeviceProperties(&prop, device) != hipSuccess ||
        hipStreamCreateWithPriority(&c->own_stream, hipStreamNonBlocking, prio_greatest) != hipSuccess ||
        hipStreamCreateWithPriority(&c->side_stream, hipStreamNonBlocking, prio_least) != hipSuccess ||
        hipEventCreateWithFlags(&c->fit_begin, hipEventDisableTiming) != hipSuccess) {
        set_error("pmk_ctx_create: device %d: %s", device, hipGetErrorString(hipGetLastError()));
        pmk_ctx_destroy(c);                 // releases whichever of the streams / events were created
        return -100;
    }
    c->num_cu = prop.multiProcessorCount;
    c->stream = c->own_stream;
    if (const char *e = std::getenv("PMK_PIPELINE_K1")) c->pipeline_k1 = std::atoi(e) != 0;      // A/B switch
    if (hipMalloc((void **)&c->d_clk, sizeof(unsigned long long) * 130 * 8) != hipSuccess ||
        hipMemset(c->d_clk, 0, sizeof(unsigned long long) * 130 * 8) != hipSuccess) {
        set_error("pmk_ctx_create: device %d: %s", device, hipGetErrorString(hipGetLastError()));
        pmk_ctx_destroy(c);
        return -100;
    }
    // kernel attributes are per device: set them for this context's device (current after hipSetDevice above)
    if (pmk::f64::set_device_attributes() || pmk::f32::set_device_attributes() || pmk::set_plan_attributes()) {
        pmk_ctx_destroy(c);
        return -100;
    }
    *out = c;
    return 0;
}

int pmk_ctx_set_stream(pmk_ctx *ctx, void *hip_stream)
{
    if (!ctx) { set_error("pmk_ctx_set_stream: ctx is NULL"); return -1; }
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return 0;
}

int pmk_ctx_set_stream_null(pmk_ctx *ctx)
{
    if (!ctx) { set_error("pmk_ctx_set_stream_null: ctx is NULL"); return -1; }
    ctx->stream = nullptr;               // the device's legacy default stream
    return 0;
}

int pmk_ctx_set_pipeline(pmk_ctx *ctx, int on)
{
    if (!ctx) { set_error("pmk_ctx_set_pipeline: ctx is NULL"); return -1; }
    ctx->pipeline_k1 = on != 0;
    return 0;
}

int pmk_ctx_synchronize(pmk_ctx *ctx)
{
    if (!ctx) { set_error("pmk_ctx_synchronize: ctx is NULL"); return -1; }
    PMK_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

void pmk_ctx_destroy(pmk_ctx *ctx)
{
    if (!ctx) return;
    for (auto &t : ctx->tm) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    for (auto &e : ctx->panel_ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
    if (ctx->fit_begin) (void)hipEventDestroy(ctx->fit_begin);
    for (auto &e : ctx->col_ev) (void)hipEventDestroy(e);
    if (ctx->d_clk) (void)hipFree(ctx->d_clk);
    delete ctx;
}

int pmk_ctx_shader_clock(pmk_ctx *ctx, int which, double *ghz)
{
    if (!ctx || !ghz || which < 0 || which > 1) { set_error("pmk_ctx_shader_clock: bad argument"); return -1; }
    PMK_HIP(hipSetDevice(ctx->device));
    unsigned long long h[130 * 8];
    PMK_HIP(hipMemcpyAsync(h, ctx->d_clk, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    PMK_HIP(hipStreamSynchronize(ctx->stream));
    double cyc = 0, ticks = 0;          // time-weighted mean over the launches and the 8 probing workgroups
    for (int x = 0; x < 8; ++x)
        for (int i = which ? 64 : 0; i < (which ? 65 : 64); ++i) {
            cyc += (double)h[130 * x + 2 * i];
            ticks += (double)h[130 * x + 2 * i + 1];
        }
    *ghz = ticks > 0 ? cyc / (ticks * 10.0) : 0.0;          // ticks of 10 ns
    return 0;
}

int pmk_ctx_enable_timers(pmk_ctx *ctx, int on)
{
    if (!ctx) { set_error("ctx is NULL"); return -1; }
    ctx->timers = on;
    return 0;
}

int pmk_ctx_timer_ms(pmk_ctx *ctx, const char *stage, double *ms)
{
    if (!ctx || !stage || !ms) { set_error("pmk_ctx_timer_ms: NULL argument"); return -1; }
    if (std::string(stage) == "panel" && ctx->panel_n > 0) {     // summed over the panel launches of the last fit
        double tot = 0;
        for (int i = 0; i < ctx->panel_n; ++i) {
            PMK_HIP(hipEventSynchronize(ctx->panel_ev[(size_t)i].second));
            float f = 0;
            PMK_HIP(hipEventElapsedTime(&f, ctx->panel_ev[(size_t)i].first, ctx->panel_ev[(size_t)i].second));
            tot += f;
        }
        *ms = tot;
        return 0;
    }
    if (std::strncmp(stage, "step:", 5) == 0) {                    // one factorisation step launch of the last fit
        const int i = atoi(stage + 5);
        if (i < 0 || i >= ctx->panel_n) { set_error("no timing recorded for stage '%s'", stage); return -2; }
        PMK_HIP(hipEventSynchronize(ctx->panel_ev[(size_t)i].second));
        float f = 0;
        PMK_HIP(hipEventElapsedTime(&f, ctx->panel_ev[(size_t)i].first, ctx->panel_ev[(size_t)i].second));
        *ms = f;
        return 0;
    }
    for (auto &t : ctx->tm)
        if (t.name == stage && t.valid) {
            PMK_HIP(hipEventSynchronize(t.b));
            float f = 0;
            PMK_HIP(hipEventElapsedTime(&f, t.a, t.b));
            *ms = f;
            return 0;
        }
    set_error("no timing recorded for stage '%s'", stage);
    return -2;
}

// ------------------------------------------------------------------------------------------ BSP
int pmk_bsp_build(int D, int64_t N, const double *X, int levels, int sign_mode, int dot_mode, pmk_bsp **out)
{
    if (!out) { set_error("pmk_bsp_build: out is NULL"); return -6; }
    *out = nullptr;
    if (D < 1 || D > MAX_D) { set_error("pmk_bsp_build: D=%d outside 1..%d", D, MAX_D); return -1; }
    if (N < 1 || !X) { set_error("pmk_bsp_build: empty point set"); return -2; }
    if (levels < 2 || levels > 31) { set_error("pmk_bsp_build: levels=%d must be in 2..31", levels); return -4; }
    if ((N >> (levels - 1)) < 1) { set_error("pmk_bsp_build: N=%lld < 2^(levels-1)", (long long)N); return -4; }
    pmk_bsp *b = new (std::nothrow) pmk_bsp();
    if (!b) { set_error("out of memory"); return -100; }
    int rc = bsp_build(D, N, X, levels, sign_mode, dot_mode != 0, b->t);
    if (rc) { delete b; return rc; }
    *out = b;
    return 0;
}

int pmk_bsp_build_device(pmk_ctx *ctx, int D, int64_t N, const double *X, int levels, int sign_mode, int dot_mode,
                         pmk_bsp **out)
{
    if (!out) { set_error("pmk_bsp_build_device: out is NULL"); return -6; }
    *out = nullptr;
    if (!ctx) { set_error("pmk_bsp_build_device: context is NULL"); return -1; }
    if (D < 1 || D > MAX_D) { set_error("pmk_bsp_build_device: D=%d outside 1..%d", D, MAX_D); return -1; }
    if (N < 1 || N >= 0x7fffffff || !X) { set_error("pmk_bsp_build_device: N must be in 1..2^31-2"); return -2; }
    if (levels < 2 || levels > 31) { set_error("pmk_bsp_build_device: levels=%d must be in 2..31", levels); return -4; }
    if ((N >> (levels - 1)) < 1) { set_error("pmk_bsp_build_device: N=%lld < 2^(levels-1)", (long long)N); return -4; }
    PMK_HIP(hipSetDevice(ctx->device));
    pmk_bsp *b = new (std::nothrow) pmk_bsp();
    if (!b) { set_error("out of memory"); return -100; }
    int rc = bsp_build_device(ctx, D, N, X, levels, sign_mode, dot_mode != 0, b->t);
    if (rc) { delete b; return rc; }
    *out = b;
    return 0;
}

int pmk_bsp_from_hyperplanes(int D, int levels, const double *hp_v, const double *hp_c, int dot_mode, pmk_bsp **out)
{
    if (!out || !hp_v || !hp_c) { set_error("pmk_bsp_from_hyperplanes: NULL argument"); return -1; }
    *out = nullptr;
    if (D < 1 || D > MAX_D || levels < 2 || levels > 31) { set_error("pmk_bsp_from_hyperplanes: bad D/levels"); return -1; }
    pmk_bsp *b = new (std::nothrow) pmk_bsp();
    if (!b) { set_error("out of memory"); return -100; }
    bsp_from_hyperplanes(D, levels, hp_v, hp_c, dot_mode != 0, b->t);
    *out = b;
    return 0;
}

void pmk_bsp_destroy(pmk_bsp *bsp) { delete bsp; }
int pmk_bsp_dim(const pmk_bsp *bsp) { return bsp ? bsp->t.D : -1; }
int pmk_bsp_levels(const pmk_bsp *bsp) { return bsp ? bsp->t.levels : -1; }
int pmk_bsp_dot_mode(const pmk_bsp *bsp) { return bsp ? bsp->t.dot_mode : -1; }
int64_t pmk_bsp_num_leaves(const pmk_bsp *bsp) { return bsp ? bsp->t.P : -1; }
int64_t pmk_bsp_num_points(const pmk_bsp *bsp) { return bsp ? bsp->t.N : -1; }

int pmk_bsp_arrays(const pmk_bsp *bsp, double *hp_v, double *hp_c, int64_t *leaf_offsets, int64_t *leaf_inds)
{
    if (!bsp) { set_error("pmk_bsp_arrays: bsp is NULL"); return -1; }
    const BspArrays &t = bsp->t;
    for (int64_t k = 0; k < t.P - 1; ++k) {
        const int64_t h = t.pre[(size_t)k];
        if (hp_v) for (int d = 0; d < t.D; ++d) hp_v[k * t.D + d] = t.v[(size_t)(h * t.D + d)];
        if (hp_c) hp_c[k] = t.c[(size_t)h];
    }
    if (leaf_offsets) std::memcpy(leaf_offsets, t.leaf_off.data(), sizeof(int64_t) * (size_t)(t.P + 1));
    if (leaf_inds && !t.leaf_inds.empty()) std::memcpy(leaf_inds, t.leaf_inds.data(), sizeof(int64_t) * t.leaf_inds.size());
    return 0;
}

int pmk_bsp_assign(const pmk_bsp *bsp, int64_t N, const double *X, double eps, int64_t *offsets, int64_t *inds,
                   int64_t *list_offsets, int64_t *lists)
{
    if (!bsp || !offsets || (N > 0 && !X)) { set_error("pmk_bsp_assign: NULL argument"); return -1; }
    return bsp_assign(bsp->t, N, X, eps, offsets, inds, list_offsets, lists);
}

int pmk_bsp_assign_device(pmk_ctx *ctx, const pmk_bsp *bsp, int64_t N, const double *X, double eps, int64_t *offsets,
                          int64_t *inds, int64_t *list_offsets, int64_t *lists)
{
    if (!ctx || !bsp || !offsets || (N > 0 && !X)) { set_error("pmk_bsp_assign_device: NULL argument"); return -1; }
    if (N < 0 || N >= 0x7fffffff) { set_error("pmk_bsp_assign_device: N must be below 2^31-1"); return -2; }
    PMK_HIP(hipSetDevice(ctx->device));
    return bsp_assign_device(ctx, bsp->t, N, X, eps, offsets, inds, list_offsets, lists);
}

int64_t pmk_bsp_findpartition(const pmk_bsp *bsp, const double *x)
{
    if (!bsp || !x) { set_error("pmk_bsp_findpartition: NULL argument"); return -1; }
    return bsp_find(bsp->t, x);
}

int64_t pmk_bsp_neighbours(const pmk_bsp *bsp, const double *p, double radius, double delta, int64_t home,
                           int64_t *region_inds, double *ts, double *zs, uint8_t *keep)
{
    if (!bsp || !p || !region_inds) { set_error("pmk_bsp_neighbours: NULL argument"); return -1; }
    return bsp_neighbours(bsp->t, p, radius, delta, home, region_inds, ts, zs, keep);
}

// ------------------------------------------------------------------------------------------ kernel matrix
int pmk_kernel_matrix(pmk_ctx *ctx, const pmk_kernel_desc *th, int D, int64_t n, const double *X, int64_t m,
                      const double *Z, double *K, int64_t ldk)
{
    if (!ctx) { set_error("pmk_kernel_matrix: ctx is NULL"); return -1; }
    if (!kernel_ok(th)) { set_error("pmk_kernel_matrix: unknown kernel family"); return -2; }
    if (D < 1 || D > MAX_D) { set_error("pmk_kernel_matrix: D=%d outside 1..%d", D, MAX_D); return -3; }
    if (n < 1 || !X || !K) { set_error("pmk_kernel_matrix: empty input"); return -4; }
    const bool sym = (Z == nullptr);
    const int64_t mc = sym ? n : m;
    if (mc < 1 || ldk < n) { set_error("pmk_kernel_matrix: bad m/ldk"); return -6; }
    PMK_HIP(hipSetDevice(ctx->device));
    std::vector<double> hx((size_t)(n * D)), hz;
    pack_soa(D, n, n, X, hx.data());
    DevTmp<double> dx, dz, dK;
    if (dx.alloc(n * D)) return -100;
    PMK_HIP(hipMemcpyAsync(dx, hx.data(), sizeof(double) * hx.size(), hipMemcpyHostToDevice, ctx->stream));
    if (!sym) {
        hz.resize((size_t)(mc * D));
        pack_soa(D, mc, mc, Z, hz.data());
        if (dz.alloc(mc * D)) return -100;
        PMK_HIP(hipMemcpyAsync(dz, hz.data(), sizeof(double) * hz.size(), hipMemcpyHostToDevice, ctx->stream));
    }
    if (dK.alloc(n * mc)) return -100;
    int rc = launch_kernel_matrix_dense(*th, D, n, dx, n, mc, sym ? dx : dz, sym ? n : mc, dK, n, sym, ctx->stream);
    if (!rc) {
        PMK_HIP(hipMemcpy2DAsync(K, sizeof(double) * ldk, dK, sizeof(double) * n, sizeof(double) * n, (size_t)mc,
                                 hipMemcpyDeviceToHost, ctx->stream));
    }
    PMK_HIP(hipStreamSynchronize(ctx->stream));     // also on failure: the temporaries must outlive the queued copies
    return rc;
}

int pmk_query_mean(pmk_ctx *ctx, const pmk_kernel_desc *th, int D, int64_t n, const double *X, const double *c,
                   int64_t Nq, const double *Xq, double *Yq)
{
    if (!ctx) { set_error("pmk_query_mean: ctx is NULL"); return -1; }
    if (!kernel_ok(th)) { set_error("pmk_query_mean: unknown kernel family"); return -2; }
    if (D < 1 || D > MAX_D) { set_error("pmk_query_mean: D=%d outside 1..%d", D, MAX_D); return -3; }
    if (n < 1 || !X || !c) { set_error("pmk_query_mean: empty model"); return -4; }
    if (Nq < 1 || !Xq || !Yq) { set_error("pmk_query_mean: empty query (the reference asserts !isempty(Xq))"); return -7; }
    PMK_HIP(hipSetDevice(ctx->device));
    std::vector<double> hx((size_t)(n * D));
    pack_soa(D, n, n, X, hx.data());
    DevTmp<double> dx, dc, dq, dy;
    if (dx.alloc(n * D) || dc.alloc(n) || dq.alloc(Nq * D) || dy.alloc(Nq)) return -100;
    PMK_HIP(hipMemcpyAsync(dx, hx.data(), sizeof(double) * hx.size(), hipMemcpyHostToDevice, ctx->stream));
    PMK_HIP(hipMemcpyAsync(dc, c, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    PMK_HIP(hipMemcpyAsync(dq, Xq, sizeof(double) * Nq * D, hipMemcpyHostToDevice, ctx->stream));
    int rc = launch_query_mean(th, 1, D, n, dx, n, dc, Nq, dq, dy, ctx->stream);
    if (!rc) PMK_HIP(hipMemcpyAsync(Yq, dy, sizeof(double) * Nq, hipMemcpyDeviceToHost, ctx->stream));
    PMK_HIP(hipStreamSynchronize(ctx->stream));
    return rc;
}

int pmk_query_mean_multi(pmk_ctx *ctx, const pmk_kernel_desc *ths, int D, int64_t n, const double *X, const double *c,
                         int64_t Nq, const double *Xq, double *Yq)
{
    if (!ctx) { set_error("pmk_query_mean_multi: ctx is NULL"); return -1; }
    if (!ths) { set_error("pmk_query_mean_multi: no kernels"); return -2; }
    if (D < 1 || D > MAX_D) { set_error("pmk_query_mean_multi: D=%d outside 1..%d", D, MAX_D); return -3; }
    if (n < 1 || !X || !c) { set_error("pmk_query_mean_multi: empty model"); return -4; }
    for (int64_t i = 0; i < n; ++i)
        if (!kernel_ok(ths + i)) { set_error("pmk_query_mean_multi: unknown kernel family at centre %lld", (long long)i); return -2; }
    if (Nq < 1 || !Xq || !Yq) { set_error("pmk_query_mean_multi: empty query (the reference asserts !isempty(Xq))"); return -7; }
    PMK_HIP(hipSetDevice(ctx->device));
    std::vector<double> hx((size_t)(n * D));
    pack_soa(D, n, n, X, hx.data());
    DevTmp<double> dx, dc, dq, dy;
    DevTmp<pmk_kernel_desc> dth;
    if (dx.alloc(n * D) || dc.alloc(n) || dq.alloc(Nq * D) || dy.alloc(Nq) || dth.alloc(n)) return -100;
    PMK_HIP(hipMemcpyAsync(dx, hx.data(), sizeof(double) * hx.size(), hipMemcpyHostToDevice, ctx->stream));
    PMK_HIP(hipMemcpyAsync(dc, c, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    PMK_HIP(hipMemcpyAsync(dq, Xq, sizeof(double) * Nq * D, hipMemcpyHostToDevice, ctx->stream));
    PMK_HIP(hipMemcpyAsync(dth, ths, sizeof(pmk_kernel_desc) * n, hipMemcpyHostToDevice, ctx->stream));
    int rc = n == 1 ? launch_query_mean(ths, 1, D, n, dx, n, dc, Nq, dq, dy, ctx->stream)
                    : launch_query_mean(dth, (int)n, D, n, dx, n, dc, Nq, dq, dy, ctx->stream);
    if (!rc) PMK_HIP(hipMemcpyAsync(Yq, dy, sizeof(double) * Nq, hipMemcpyDeviceToHost, ctx->stream));
    PMK_HIP(hipStreamSynchronize(ctx->stream));
    return rc;
}

// ------------------------------------------------------------------------------------------ model
void pmk_model_destroy(pmk_model *m)
{
    if (!m) return;
    dev_free(m->d_desc); dev_free(m->d_info); dev_free(m->d_hv); dev_free(m->d_hc); dev_free(m->d_pre);
    dev_free(m->d_order);
    dev_free(m->d_sched);
    dev_free(m->d_sched_init);
    for (void **p : {&m->d_qtasks, &m->d_diag, &m->d_x, &m->d_y, &m->d_z, &m->d_c, &m->d_a, &m->d_inv, &m->d_strip, &m->d_partial, &m->d_solve_part, &m->d_chain}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    delete m;
}

static int upload_targets(pmk_model *m, const double *const *y)
{
    std::vector<double> hy((size_t)m->tot_y, 0.0);
    for (int64_t r = 0; r < m->P; ++r) {
        if (!y[r]) { set_error("targets of patch %lld are NULL", (long long)r); return -6; }
        std::memcpy(hy.data() + m->desc[(size_t)r].yoff, y[r], sizeof(double) * (size_t)m->desc[(size_t)r].n);
    }
    return upload_real(m, m->d_y, 0, hy.data(), hy.size());
}

int pmk_model_create(pmk_ctx *ctx, int D, int64_t P, const int64_t *n, const double *const *X,
                     const double *const *y, pmk_model **out)
{
    return pmk_model_create_ex(ctx, D, P, n, X, y, PMK_F64, out);
}

int pmk_model_create_ex(pmk_ctx *ctx, int D, int64_t P, const int64_t *n, const double *const *X,
                        const double *const *y, int dtype, pmk_model **out)
{
    if (!out) { set_error("pmk_model_create: out is NULL"); return -7; }
    *out = nullptr;
    if (dtype != PMK_F64 && dtype != PMK_F32) { set_error("pmk_model_create: unknown dtype %d", dtype); return -8; }
    if (!ctx) { set_error("pmk_model_create: ctx is NULL"); return -1; }
    if (D < 1 || D > MAX_D) { set_error("pmk_model_create: D=%d outside 1..%d", D, MAX_D); return -2; }
    if (P < 1 || !n || !X || !y) { set_error("pmk_model_create: no patches"); return -3; }
    PMK_HIP(hipSetDevice(ctx->device));
    pmk_model *m = new (std::nothrow) pmk_model();
    if (!m) { set_error("out of memory"); return -100; }
    m->ctx = ctx; m->D = D; m->P = P;
    m->dtype = dtype; m->esz = dtype == PMK_F32 ? 4 : 8;
    m->desc.resize((size_t)P);
    int64_t a = 0, xo = 0, yo = 0, io = 0;
    for (int64_t r = 0; r < P; ++r) {
        if (n[r] < 1 || n[r] > (1 << 24) || !X[r]) {
            set_error("pmk_model_create: patch %lld has n=%lld (the reference asserts a non-empty patch)", (long long)r,
                      (long long)n[r]);
            delete m;
            return -4;
        }
        PatchDesc &d = m->desc[(size_t)r];
        d.n = (int32_t)n[r];
        d.nt = (int32_t)((n[r] + TILE - 1) / TILE);
        d.ld = d.nt * TILE;
        d.pad_ = 0;
        d.aoff = a; d.xoff = xo; d.yoff = yo; d.ioff = io;
        a += (int64_t)d.ld * d.ld;
        xo += (int64_t)d.ld * D;
        yo += d.ld;
        io += (int64_t)d.nt * 4 * 32 * 32;
        m->max_nt = std::max(m->max_nt, (int)d.nt);
    }
    m->tot_a = a; m->tot_x = xo; m->tot_y = yo; m->tot_inv = io;
    int rc = 0;
    rc |= dev_alloc(&m->d_desc, P);
    auto alloc_real = [&](void **p, int64_t count) {
        *p = nullptr;
        return hipMalloc(p, m->esz * (size_t)std::max<int64_t>(count, 1)) == hipSuccess ? 0 : -100;
    };
    rc |= alloc_real(&m->d_x, xo);
    rc |= alloc_real(&m->d_y, yo);
    rc |= alloc_real(&m->d_z, yo);
    rc |= alloc_real(&m->d_c, yo);
    rc |= alloc_real(&m->d_a, a);
    rc |= alloc_real(&m->d_inv, io);
    rc |= dev_alloc(&m->d_info, P);
    rc |= dev_alloc(&m->d_order, P);
    if (rc) { pmk_model_destroy(m); return -100; }
    {
        // factorisation order: by tile count, largest first (stable, so equal sizes keep the caller's order)
        std::vector<int32_t> order((size_t)P);
        for (int64_t r = 0; r < P; ++r) order[(size_t)r] = (int32_t)r;
        std::stable_sort(order.begin(), order.end(),
                         [&](int32_t a2, int32_t b2) { return m->desc[(size_t)a2].nt > m->desc[(size_t)b2].nt; });
        // one workgroup per block row fills the chip only if there are enough patches: P (max_nt - 1) / 2 block rows per
        // step on average against two workgroups per CU.  Below that -- single large problems, fitRKHS! at scale -- the
        // factorisation takes the split path (pmk_chol.hip).  The two paths sum in different orders (last-bit differences
        // in L), so the choice is kept away from everyday batches: only patches of >= 32 tiles (n > 3968) qualify, and a
        // model and its shards -- which hold the same patch sizes -- then decide alike unless they straddle P's bound.
        m->split_mode = m->max_nt >= 32 && P * (int64_t)(m->max_nt - 1) / 2 < 2 * (int64_t)ctx->num_cu;
        m->active_prefix.assign((size_t)m->max_nt + 2, 0);
        for (int64_t r = 0; r < P; ++r)
            for (int t = 0; t <= m->desc[(size_t)r].nt; ++t) ++m->active_prefix[(size_t)t];
        m->order = order;
        if (const char *e = std::getenv("PMK_CHOL_QUEUE")) m->queue_mode = std::atoi(e) != 0;     // A/B switches
        if (const char *e = std::getenv("PMK_QUEUE_FROM")) m->queue_from = std::atoi(e);
        if (hipMemcpy(m->d_order, order.data(), sizeof(int32_t) * (size_t)P, hipMemcpyHostToDevice) != hipSuccess) {
            set_error("pmk_model_create: upload failed");
            pmk_model_destroy(m);
            return -100;
        }
    }
    {
        std::vector<double> hx((size_t)xo);
        for (int64_t r = 0; r < P; ++r) {
            const PatchDesc &d = m->desc[(size_t)r];
            pack_soa(D, d.n, d.ld, X[r], hx.data() + d.xoff);
        }
        if (upload_real(m, m->d_x, 0, hx.data(), hx.size()) ||
            hipMemcpy(m->d_desc, m->desc.data(), sizeof(PatchDesc) * (size_t)P, hipMemcpyHostToDevice) != hipSuccess) {
            set_error("pmk_model_create: upload failed");
            pmk_model_destroy(m);
            return -100;
        }
    }
    rc = upload_targets(m, y);
    if (rc) { pmk_model_destroy(m); return rc; }
    // on the context's (non-blocking) stream: a null-stream memset would not be ordered with the fit that follows
    if (hipMemsetAsync(m->d_info, 0, sizeof(int32_t) * (size_t)P, ctx->stream) != hipSuccess) {
        set_error("pmk_model_create: memset failed");
        pmk_model_destroy(m);
        return -100;
    }
    *out = m;
    return 0;
}

int pmk_model_set_diag(pmk_model *m, const double *const *diag)
{
    if (!m) { set_error("pmk_model_set_diag: model is NULL"); return -1; }
    PMK_HIP(hipSetDevice(m->ctx->device));
    PMK_HIP(hipStreamSynchronize(m->ctx->stream));
    m->fitted = false;
    if (!diag) {
        if (m->d_diag) (void)hipFree(m->d_diag);
        m->d_diag = nullptr;
        return 0;
    }
    if (!m->d_diag && hipMalloc(&m->d_diag, m->esz * (size_t)std::max<int64_t>(m->tot_y, 1)) != hipSuccess) {
        set_error("pmk_model_set_diag: out of device memory");
        return -100;
    }
    std::vector<double> hd((size_t)m->tot_y, 0.0);
    for (int64_t r = 0; r < m->P; ++r) {
        if (!diag[r]) { set_error("pmk_model_set_diag: the addends of patch %lld are NULL", (long long)r); return -2; }
        std::memcpy(hd.data() + m->desc[(size_t)r].yoff, diag[r], sizeof(double) * (size_t)m->desc[(size_t)r].n);
    }
    return upload_real(m, m->d_diag, 0, hd.data(), hd.size());
}

int pmk_query_set_diag(pmk_query *q, const double *diag)
{
    if (!q) { set_error("pmk_query_set_diag: query is NULL"); return -1; }
    pmk_ctx *c = q->m->ctx;
    PMK_HIP(hipSetDevice(c->device));
    PMK_HIP(hipStreamSynchronize(c->stream));
    dev_free(q->d_qdiag);
    if (!diag || q->Nq == 0) return 0;
    if (dev_alloc(&q->d_qdiag, q->Nq)) return -100;
    PMK_HIP(hipMemcpy(q->d_qdiag, diag, sizeof(double) * (size_t)q->Nq, hipMemcpyHostToDevice));
    return 0;
}

int pmk_model_set_targets(pmk_model *m, const double *const *y)
{
    if (!m || !y) { set_error("pmk_model_set_targets: NULL argument"); return -1; }
    PMK_HIP(hipSetDevice(m->ctx->device));
    PMK_HIP(hipStreamSynchronize(m->ctx->stream));
    m->fitted = false;
    return upload_targets(m, y);
}

/* include/pmk_test.h: force the factorisation path of a model (tests compare the two on the same data) */
int pmk_test_model_set_split(pmk_model *m, int on)
{
    if (!m) { set_error("pmk_test_model_set_split: model is NULL"); return -1; }
    m->split_mode = on != 0 && m->max_nt >= 2;
    m->chain_mode = on == 2 ? 0 : on == 3 ? 1 : -1;
    return 0;
}

int pmk_model_fit(pmk_model *m, const pmk_kernel_desc *th, double sigma2)
{
    if (!m) { set_error("pmk_model_fit: model is NULL"); return -1; }
    if (!kernel_ok(th)) { set_error("pmk_model_fit: unknown kernel family"); return -2; }
    PMK_HIP(hipSetDevice(m->ctx->device));
    pmk_ctx *c = m->ctx;
    m->th = *th;
    m->sigma2 = sigma2;
    int rc;
    c->tic("fit");
    // Pipelined kernel-matrix build (OFF by default: measured 0.3 ms SLOWER at config C, profiles/r02_fit_experiments.txt;
    // pmk_ctx_set_pipeline turns it on; never while stage timers are on or on the split path): K1 is HBM-write bound
    // and the step launches MFMA bound, so K1 goes block column by block column onto a low-priority side stream and
    // launch l of the factorisation waits for the columns it reads (stages <= l + 1).  The side stream starts behind
    // everything already queued on the context's stream (the slabs still hold the previous factor, which earlier
    // predictions may be reading).
    const bool pipelined = c->pipeline_k1 && !c->timers && !m->split_mode && m->max_nt >= 3 && c->side_stream;
    m->fuse_k1 = false;
    if (pipelined) {
        const int n_ev = m->max_nt;
        while ((int)c->col_ev.size() < n_ev) {
            hipEvent_t e;
            PMK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            c->col_ev.push_back(e);
        }
        PMK_HIP(hipEventRecord(c->fit_begin, c->stream));
        PMK_HIP(hipStreamWaitEvent(c->side_stream, c->fit_begin, 0));
        for (int st = 0; st < n_ev; ++st) {
            if ((rc = PMK_BY_DTYPE(m, launch_kernel_matrix_slabs(m, *th, sigma2, c->side_stream, 0, m->P, st)))) return rc;
            PMK_HIP(hipEventRecord(c->col_ev[(size_t)st], c->side_stream));
        }
        if ((rc = PMK_BY_DTYPE(m, launch_cholesky(m, c->stream, 0, m->P, c->col_ev.data(), n_ev)))) return rc;
    } else {
        // Fused kernel-matrix build (PMK_FUSE_K1=0 turns it off): for the compact Spline34 profile in 2 or 3 dimensions K1
        // writes the diagonal 128 x 128 tiles only, and the factorisation's step launches evaluate every tile below them at
        // its one use instead of reading it from the slab (same kern_eval: the same bits).  Not on the split or queue paths.
        const char *fe = std::getenv("PMK_FUSE_K1");
        const bool fuse_env = !(fe && std::atoi(fe) == 0);
        m->fuse_k1 = fuse_env && th->family == PMK_SPLINE34 && (m->D == 2 || m->D == 3) && !m->split_mode && !m->queue_mode &&
                     m->max_nt >= 2;
        c->tic("kernel_matrix");
        if ((rc = PMK_BY_DTYPE(m, launch_kernel_matrix_slabs(m, *th, sigma2, c->stream, 0, m->P, m->fuse_k1 ? -2 : -1)))) return rc;
        c->toc("kernel_matrix");
        c->tic("cholesky");
        if ((rc = PMK_BY_DTYPE(m, launch_cholesky(m, c->stream, 0, m->P, nullptr, 0)))) return rc;
        c->toc("cholesky");
    }
    c->tic("solve");
    if ((rc = PMK_BY_DTYPE(m, launch_backsolve(m, c->stream, 0, m->P)))) return rc;
    c->toc("solve");
    c->toc("fit");
    m->fitted = true;
    return 0;
}

int pmk_model_info(pmk_model *m, int32_t *info)
{
    if (!m || !info) { set_error("pmk_model_info: NULL argument"); return -1; }
    PMK_HIP(hipSetDevice(m->ctx->device));
    PMK_HIP(hipMemcpyAsync(info, m->d_info, sizeof(int32_t) * (size_t)m->P, hipMemcpyDeviceToHost, m->ctx->stream));
    constexpr int SCHED_HEADS = 16;                 // pmk_chol.hip: error word, then 8 list heads per segment
    std::vector<int32_t> sched((size_t)SCHED_HEADS + 8 * 16, 0);
    if (m->queue_used)
        PMK_HIP(hipMemcpyAsync(sched.data(), m->d_sched, sizeof(int32_t) * sched.size(), hipMemcpyDeviceToHost, m->ctx->stream));
    int32_t chain_err = 0;
    if (m->chain_used)
        PMK_HIP(hipMemcpyAsync(&chain_err, m->d_chain, sizeof(int32_t), hipMemcpyDeviceToHost, m->ctx->stream));
    PMK_HIP(hipStreamSynchronize(m->ctx->stream));
    if (chain_err != 0) {
        // solve_chain_kernel: a block waited 3 s for the block it depends on (never seen; bounded so that a fault ends)
        PMK_HIP(hipMemsetAsync(m->d_chain, 0, sizeof(int32_t), m->ctx->stream));
        set_error("pmk_model_fit: the chained back substitution timed out waiting for block %d", (int)chain_err - 1);
        return -4;
    }
    if (m->queue_used) {
        // the task-queue factorisation: no time-out, and every XCD's list drained
        bool drained = sched[0] == 0;
        for (int sg = 0; sg < m->qsegs; ++sg)
            for (int x = 0; x < 8; ++x)
                drained = drained && sched[(size_t)SCHED_HEADS + 8 * sg + x] >= m->qoff[(size_t)sg * 9 + x + 1] - m->qoff[(size_t)sg * 9 + x];
        if (!drained) {
            if (std::getenv("PMK_QUEUE_DEBUG")) {
                std::fprintf(stderr, "queue: err %d", sched[0]);
                for (int x = 0; x < 8; ++x)
                    std::fprintf(stderr, " | head %d of %d", sched[(size_t)SCHED_HEADS + x], m->qoff[(size_t)x + 1] - m->qoff[(size_t)x]);
                std::fprintf(stderr, "\n");
                std::vector<int32_t> fl((size_t)m->P * (size_t)m->qfstride);
                if (hipMemcpy(fl.data(), m->d_sched + 16 + 8 * 16, sizeof(int32_t) * fl.size(), hipMemcpyDeviceToHost) == hipSuccess)
                    for (int64_t r = 0; r < m->P; ++r) {
                        const int32_t *f = fl.data() + (size_t)r * (size_t)m->qfstride;
                        if (f[0] >= m->desc[(size_t)r].nt) continue;
                        std::fprintf(stderr, "patch %lld nt %d: diag %d look %d rows", (long long)r, m->desc[(size_t)r].nt, f[0], f[m->qfstride - 1]);
                        for (int i = 0; i < m->desc[(size_t)r].nt; ++i) std::fprintf(stderr, " %d", f[1 + i]);
                        std::fprintf(stderr, "\n");
                    }
            }
            set_error("pmk_model_fit: the factorisation's task queue did not complete (error word %d)", sched[0]);
            return -101;
        }
    }
    int worst = 0;
    for (int64_t r = 0; r < m->P; ++r) {
        // a failure inside the identity padding cannot happen; clamp to the patch size for safety
        if (info[r] > m->desc[(size_t)r].n) info[r] = m->desc[(size_t)r].n;
        if (info[r] > 0 && !worst) worst = 1;
    }
    return worst;
}

int64_t pmk_model_num_patches(const pmk_model *m) { return m ? m->P : -1; }

int pmk_model_get(pmk_model *m, int64_t patch, int what, double *out, int64_t ld)
{
    if (!m || !out) { set_error("pmk_model_get: NULL argument"); return -1; }
    if (patch < 0 || patch >= m->P) { set_error("pmk_model_get: patch %lld of %lld", (long long)patch, (long long)m->P); return -2; }
    const PatchDesc &d = m->desc[(size_t)patch];
    pmk_ctx *c = m->ctx;
    PMK_HIP(hipSetDevice(c->device));
    if (what != PMK_GET_K && !m->fitted) { set_error("pmk_model_get: model is not fitted"); return -3; }
    switch (what) {
    case PMK_GET_C:
        return download_real_2d(m, out, d.n, m->d_c, d.yoff, d.ld, d.n, 1, c->stream);
    case PMK_GET_L: {
        if (ld < d.n) { set_error("pmk_model_get: ld too small"); return -5; }
        if (int rc = download_real_2d(m, out, ld, m->d_a, d.aoff, d.ld, d.n, d.n, c->stream)) return rc;
        for (int64_t j = 1; j < d.n; ++j)
            for (int64_t i = 0; i < j; ++i) out[i + j * ld] = 0.0;   // .L of the reference: strict upper = 0
        return 0;
    }
    case PMK_GET_K: {
        // U_set entry (mixtureGP.jl:99): K without noise, rebuilt on demand from the resident points
        if (ld < d.n) { set_error("pmk_model_get: ld too small"); return -5; }
        if (!kernel_ok(&m->th)) { set_error("pmk_model_get: no kernel set (fit first)"); return -3; }
        DevTmp<double> dK, dxs;
        if (dK.alloc((int64_t)d.n * d.n) || dxs.alloc((int64_t)d.ld * m->D)) return -100;
        {   // the dense host-API kernel is fp64: give it fp64 coordinates whatever the model's element type
            std::vector<double> hx((size_t)(d.ld * m->D));
            if (int rc2 = download_real_2d(m, hx.data(), d.ld, m->d_x, d.xoff, d.ld, d.ld, m->D, c->stream)) return rc2;
            PMK_HIP(hipMemcpy(dxs, hx.data(), sizeof(double) * hx.size(), hipMemcpyHostToDevice));
        }
        int rc = launch_kernel_matrix_dense(m->th, m->D, d.n, dxs, d.ld, d.n, dxs, d.ld, dK, d.n, true, c->stream);
        if (!rc)
            PMK_HIP(hipMemcpy2DAsync(out, sizeof(double) * ld, dK, sizeof(double) * d.n, sizeof(double) * d.n, (size_t)d.n,
                                     hipMemcpyDeviceToHost, c->stream));
        PMK_HIP(hipStreamSynchronize(c->stream));
        if (!rc && m->d_diag) {                   // the kernel's own diagonal term is part of K (not the noise)
            std::vector<double> hd((size_t)d.n);
            if (int rc2 = download_real_2d(m, hd.data(), d.n, m->d_diag, d.yoff, d.ld, d.n, 1, c->stream)) return rc2;
            for (int64_t i = 0; i < d.n; ++i) out[i + i * ld] += hd[(size_t)i];
        }
        return rc;
    }
    case PMK_GET_LINV_DIAG:
        return download_real_2d(m, out, (int64_t)d.nt * 4096, m->d_inv, d.ioff, (int64_t)d.nt * 4096, (int64_t)d.nt * 4096, 1,
                                c->stream);
    default:
        set_error("pmk_model_get: unknown selector %d", what);
        return -3;
    }
}

int pmk_fit_batched(pmk_ctx *ctx, const pmk_kernel_desc *th, double sigma2, int D, int64_t P, const int64_t *n,
                    const double *const *X, const double *const *y, pmk_model **out, double *const *c_out,
                    int32_t *info)
{
    if (!out) { set_error("pmk_fit_batched: out is NULL"); return -9; }
    int rc = pmk_model_create(ctx, D, P, n, X, y, out);
    if (rc) return rc;
    rc = pmk_model_fit(*out, th, sigma2);
    if (rc) { pmk_model_destroy(*out); *out = nullptr; return rc; }
    std::vector<int32_t> tmp((size_t)P);
    int st = pmk_model_info(*out, info ? info : tmp.data());
    if (st < 0) return st;
    if (c_out) {
        bool every = true;
        for (int64_t r = 0; r < P; ++r) every = every && c_out[r] != nullptr;
        if (every) {
            if ((rc = pmk_model_get_weights(*out, c_out))) return rc;
        } else {
            for (int64_t r = 0; r < P; ++r)
                if (c_out[r] && (rc = pmk_model_get(*out, r, PMK_GET_C, c_out[r], 0))) return rc;
        }
    }
    return st;
}

int pmk_model_load(pmk_ctx *ctx, int D, int64_t P, const int64_t *n, const double *const *X, const double *const *c,
                   const double *const *L, const int64_t *ldl, pmk_model **out)
{
    if (!out) { set_error("pmk_model_load: out is NULL"); return -9; }
    *out = nullptr;
    if (!c || !L || !ldl) { set_error("pmk_model_load: NULL factors"); return -6; }
    // geometry + coordinates through the ordinary constructor (targets are not needed: pass c as a stand-in)
    int rc = pmk_model_create(ctx, D, P, n, X, c, out);
    if (rc) return rc;
    pmk_model *m = *out;
    std::vector<double> slab;
    for (int64_t r = 0; r < P; ++r) {
        const PatchDesc &d = m->desc[(size_t)r];
        if (!L[r] || !c[r] || ldl[r] < d.n) { set_error("pmk_model_load: bad factor of patch %lld", (long long)r); pmk_model_destroy(m); *out = nullptr; return -7; }
        slab.assign((size_t)d.ld * d.ld, 0.0);
        for (int64_t j = 0; j < d.ld; ++j) {
            if (j < d.n) for (int64_t i = j; i < d.n; ++i) slab[(size_t)(i + j * d.ld)] = L[r][i + j * ldl[r]];
            else slab[(size_t)(j + j * d.ld)] = 1.0;                      // identity padding
        }
        std::vector<double> cc((size_t)d.ld, 0.0);
        std::memcpy(cc.data(), c[r], sizeof(double) * (size_t)d.n);
        if ((rc = upload_real(m, m->d_a, d.aoff, slab.data(), slab.size())) ||
            (rc = upload_real(m, m->d_c, d.yoff, cc.data(), cc.size()))) {
            pmk_model_destroy(m);
            *out = nullptr;
            return rc;
        }
    }
    if (!(rc = PMK_BY_DTYPE(m, launch_ninv_from_slabs(m, ctx->stream))) && hipStreamSynchronize(ctx->stream) != hipSuccess) {
        set_error("pmk_model_load: %s", hipGetErrorString(hipGetLastError()));
        rc = -100;
    }
    if (rc) { pmk_model_destroy(m); *out = nullptr; return rc; }
    m->fitted = true;
    return 0;
}

int pmk_model_set_weights(pmk_model *m, const double *const *c)
{
    if (!m || !c) { set_error("pmk_model_set_weights: NULL argument"); return -1; }
    if (!m->fitted) { set_error("pmk_model_set_weights: model is not fitted"); return -3; }
    PMK_HIP(hipSetDevice(m->ctx->device));
    PMK_HIP(hipStreamSynchronize(m->ctx->stream));
    for (int64_t r = 0; r < m->P; ++r) {
        if (!c[r]) { set_error("pmk_model_set_weights: weights of patch %lld are NULL", (long long)r); return -2; }
        const PatchDesc &d = m->desc[(size_t)r];
        std::vector<double> cc((size_t)d.ld, 0.0);
        std::memcpy(cc.data(), c[r], sizeof(double) * (size_t)d.n);
        if (int rc = upload_real(m, m->d_c, d.yoff, cc.data(), cc.size())) return rc;
    }
    return 0;
}

int pmk_model_get_weights(pmk_model *m, double *const *c)
{
    if (!m || !c) { set_error("pmk_model_get_weights: NULL argument"); return -1; }
    if (!m->fitted) { set_error("pmk_model_get_weights: model is not fitted"); return -3; }
    PMK_HIP(hipSetDevice(m->ctx->device));
    // one transfer of the padded vector, then the patches' parts are cut out on the host
    std::vector<double> all((size_t)std::max<int64_t>(m->tot_y, 1));
    if (int rc = download_real_2d(m, all.data(), m->tot_y, m->d_c, 0, m->tot_y, m->tot_y, 1, m->ctx->stream)) return rc;
    for (int64_t r = 0; r < m->P; ++r) {
        const PatchDesc &d = m->desc[(size_t)r];
        if (!c[r]) { set_error("pmk_model_get_weights: output of patch %lld is NULL", (long long)r); return -2; }
        std::memcpy(c[r], all.data() + d.yoff, sizeof(double) * (size_t)d.n);
    }
    return 0;
}

int pmk_model_queryinner(pmk_model *m, int64_t patch, const pmk_kernel_desc *th, int64_t Nq, const double *Xq,
                         double *mu, double *var)
{
    return pmk_model_queryinner_ex(m, patch, th, Nq, Xq, 1e-12, mu, var);
}

int pmk_model_queryinner_ex(pmk_model *m, int64_t patch, const pmk_kernel_desc *th, int64_t Nq, const double *Xq,
                            double min_v, double *mu, double *var)
{
    if (!m || !m->fitted) { set_error("pmk_model_queryinner: model is not fitted"); return -1; }
    if (patch < 0 || patch >= m->P) { set_error("pmk_model_queryinner: patch %lld of %lld", (long long)patch, (long long)m->P); return -2; }
    if (!kernel_ok(th)) { set_error("pmk_model_queryinner: unknown kernel family"); return -3; }
    if (Nq < 1 || Nq > 0x7fffffff || !Xq || !mu || !var) { set_error("pmk_model_queryinner: empty query"); return -4; }
    pmk_ctx *c = m->ctx;
    PMK_HIP(hipSetDevice(c->device));
    // a plan by hand: every query is one item of region `patch`, already "sorted"
    pmk_query q;
    q.m = m; q.Nq = Nq; q.total = Nq; q.min_v = min_v;
    int rc = 0;
    rc |= dev_alloc(&q.d_xq, Nq * m->D);
    rc |= dev_alloc(&q.d_sorted_item, Nq);
    rc |= dev_alloc(&q.d_u, Nq);
    rc |= dev_alloc(&q.d_v, Nq);
    if (!rc) {
        q.d_item_query = q.d_sorted_item;            // both are the identity permutation
        q.roff.assign((size_t)(m->leaf_base + m->P + 1), 0);
        for (int64_t r = m->leaf_base + patch + 1; r <= m->leaf_base + m->P; ++r) q.roff[(size_t)r] = Nq;
        if (hipMemcpyAsync(q.d_xq, Xq, sizeof(double) * (size_t)(Nq * m->D), hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = -100;
        if (!rc) rc = launch_iota(q.d_sorted_item, Nq, c->stream);
        if (!rc) rc = PMK_BY_DTYPE(m, build_strip_tasks(&q, c->stream));
        if (!rc) rc = PMK_BY_DTYPE(m, launch_items(&q, *th, c->stream));
        if (!rc && (hipMemcpyAsync(mu, q.d_u, sizeof(double) * (size_t)Nq, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                    hipMemcpyAsync(var, q.d_v, sizeof(double) * (size_t)Nq, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                    hipStreamSynchronize(c->stream) != hipSuccess)) {
            set_error("pmk_model_queryinner: copy back failed");
            rc = -100;
        }
    } else {
        rc = -100;
    }
    dev_free(q.d_xq); dev_free(q.d_sorted_item); dev_free(q.d_u); dev_free(q.d_v);
    if (q.d_tasks) (void)hipFree(q.d_tasks);
    if (q.d_sync) (void)hipFree(q.d_sync);
    q.d_item_query = nullptr;
    return rc;
}

// ------------------------------------------------------------------------------------------ predict
int pmk_model_set_bsp(pmk_model *m, const pmk_bsp *bsp, int64_t leaf_base)
{
    if (!m || !bsp) { set_error("pmk_model_set_bsp: NULL argument"); return -1; }
    const BspArrays &t = bsp->t;
    // The tree may live in the first t.D of the model's D coordinates (warp-feature kernels: the partition is built on the
    // positions, the kernel runs on positions + appended warp values).  The normals are padded with zeros: v . x, the foot
    // points z = p + t v and their distances then come out bit for bit as in t.D dimensions (the extra products are exact
    // zeros), so leaf ids, neighbour lists and t values are those of the tree's own space.
    if (t.D > m->D) { set_error("pmk_model_set_bsp: tree dimension %d > model dimension %d", t.D, m->D); return -2; }
    if (leaf_base < 0 || leaf_base + m->P > t.P) {
        set_error("pmk_model_set_bsp: leaves [%lld, %lld) outside the tree's %lld leaves", (long long)leaf_base,
                  (long long)(leaf_base + m->P), (long long)t.P);
        return -3;
    }
    PMK_HIP(hipSetDevice(m->ctx->device));
    dev_free(m->d_hv); dev_free(m->d_hc); dev_free(m->d_pre);
    if (dev_alloc(&m->d_hv, (t.P - 1) * m->D) || dev_alloc(&m->d_hc, t.P - 1) || dev_alloc(&m->d_pre, t.P - 1)) return -100;
    std::vector<int32_t> pre32((size_t)(t.P - 1));
    for (size_t i = 0; i < pre32.size(); ++i) pre32[i] = (int32_t)t.pre[i];
    if (t.P > 1) {
        std::vector<double> hv((size_t)((t.P - 1) * m->D), 0.0);
        for (int64_t h = 0; h < t.P - 1; ++h)
            for (int d = 0; d < t.D; ++d) hv[(size_t)(h * m->D + d)] = t.v[(size_t)(h * t.D + d)];
        PMK_HIP(hipMemcpy(m->d_hv, hv.data(), sizeof(double) * hv.size(), hipMemcpyHostToDevice));
        PMK_HIP(hipMemcpy(m->d_hc, t.c.data(), sizeof(double) * t.c.size(), hipMemcpyHostToDevice));
        PMK_HIP(hipMemcpy(m->d_pre, pre32.data(), sizeof(int32_t) * pre32.size(), hipMemcpyHostToDevice));
    }
    m->levels = t.levels;
    m->dot_mode = t.dot_mode;
    m->P_global = t.P;
    m->leaf_base = leaf_base;
    return 0;
}

void pmk_query_destroy(pmk_query *q)
{
    if (!q) return;
    // d_xq and d_item_t are the bases of the two arenas (query_reserve, grow_item_buffers): the other per-point and
    // per-item pointers live inside them
    dev_free(q->d_xq); dev_free(q->d_item_t); dev_free(q->d_qdiag); dev_free(q->d_roff); dev_free(q->d_flag);
    if (q->d_tmp) (void)hipFree(q->d_tmp);
    if (q->d_sort_scratch) (void)hipFree(q->d_sort_scratch);
    if (q->d_tasks) (void)hipFree(q->d_tasks);
    if (q->d_sync) (void)hipFree(q->d_sync);
    delete q;
}

}  // extern "C"

namespace pmk {

// One device allocation cut into 256-byte aligned pieces: a query object used to own two dozen small buffers, and every
// hipFree is a device synchronisation (pmk_query_destroy: 3.4 ms of a 108 ms querymixtureGP! at config C).
struct ArenaLayout {
    size_t bytes = 0;
    size_t add(size_t n) { const size_t at = bytes; bytes = (bytes + n + 255) & ~(size_t)255; return at; }
};

// per-query-point buffers for up to Nq points, grow only.  ONE allocation; d_xq is its base (the only pointer freed).
int query_reserve(pmk_query *q, int64_t Nq)
{
    pmk_model *m = q->m;
    if (Nq <= q->nq_cap) return 0;
    const bool regrow = q->nq_cap > 0;
    dev_free(q->d_xq);
    q->d_home = nullptr; q->d_cnt = nullptr; q->d_qoff = nullptr; q->d_yq = nullptr; q->d_vq = nullptr;
    q->d_stage_r = nullptr; q->d_stage_t = nullptr;
    q->nq_cap = 0;
    const size_t cap = (size_t)(Nq + (regrow ? Nq / 8 : 0)), c1 = std::max<size_t>(cap, 1);
    ArenaLayout a;
    const size_t o_xq = a.add(sizeof(double) * c1 * (size_t)m->D), o_st = a.add(sizeof(double) * 4 * c1),      // PLAN_STAGE rows
                 o_yq = a.add(sizeof(double) * c1), o_vq = a.add(sizeof(double) * c1), o_qoff = a.add(sizeof(int64_t) * (c1 + 1)),
                 o_home = a.add(sizeof(int32_t) * c1), o_cnt = a.add(sizeof(int32_t) * (c1 + 1)), o_sr = a.add(sizeof(int32_t) * 4 * c1);
    char *base = nullptr;
    PMK_HIP(hipMalloc((void **)&base, a.bytes));
    q->d_xq = reinterpret_cast<double *>(base + o_xq);                 // o_xq == 0
    q->d_stage_t = reinterpret_cast<double *>(base + o_st);
    q->d_yq = reinterpret_cast<double *>(base + o_yq);
    q->d_vq = reinterpret_cast<double *>(base + o_vq);
    q->d_qoff = reinterpret_cast<int64_t *>(base + o_qoff);
    q->d_home = reinterpret_cast<int32_t *>(base + o_home);
    q->d_cnt = reinterpret_cast<int32_t *>(base + o_cnt);
    q->d_stage_r = reinterpret_cast<int32_t *>(base + o_sr);
    q->nq_cap = (int64_t)cap;
    return 0;
}

int grow_item_buffers(pmk_query *q, int64_t total);

// (re)load a query object with n explicit (point, region) items, one per point (host or device pointers): what
// pmk_query_create_items does after allocating; reuses the object's buffers.  Blocks (region offsets come back).
int query_set_items(pmk_query *q, int64_t n, const double *xq, const int32_t *region)
{
    pmk_model *m = q->m;
    pmk_ctx *c = m->ctx;
    hipStream_t s = c->stream;
    int rc;
    if ((rc = query_reserve(q, n))) return rc;
    q->Nq = n;
    q->total = n;
    q->planned = false;
    q->roff.assign((size_t)(m->P_global + 1), 0);
    q->ntasks = 0;
    if (n > 0) {
        int bad = 0;
        if ((rc = grow_item_buffers(q, n))) return rc;
        if (!q->d_flag && dev_alloc(&q->d_flag, 1)) return -100;
        hipError_t e = hipMemcpyAsync(q->d_xq, xq, sizeof(double) * (size_t)(n * m->D), hipMemcpyDefault, s);
        if (e == hipSuccess) e = hipMemcpyAsync(q->d_item_region, region, sizeof(int32_t) * (size_t)n, hipMemcpyDefault, s);
        if (e == hipSuccess) e = hipMemsetAsync(q->d_flag, 0, sizeof(int), s);
        if (e == hipSuccess) rc = launch_explicit_items(q, q->d_flag, s);
        if (e == hipSuccess && !rc) rc = launch_sort_items(q, s);
        if (e == hipSuccess && !rc) e = hipMemcpyAsync(&bad, q->d_flag, sizeof(int), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && !rc)
            e = hipMemcpyAsync(q->roff.data(), q->d_roff, sizeof(int64_t) * q->roff.size(), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && !rc) e = hipStreamSynchronize(s);
        if (e != hipSuccess) { set_error("query_set_items: %s", hipGetErrorString(e)); return -100; }
        if (rc) return rc;
        if (bad) {
            set_error("pmk_query_create_items: a region is outside this model's leaves [%lld, %lld)",
                      (long long)m->leaf_base, (long long)(m->leaf_base + m->P));
            return -3;
        }
        if ((rc = PMK_BY_DTYPE(m, build_strip_tasks(q, s)))) return rc;
    }
    q->planned = true;
    return 0;
}

// per-item buffers, grow only: repeated plans of one query batch reuse them.  ONE allocation; d_item_t is its base.
int grow_item_buffers(pmk_query *q, int64_t total)
{
    if (total <= q->item_cap) return 0;
    dev_free(q->d_item_t);
    q->d_item_region = nullptr; q->d_item_query = nullptr; q->d_sorted_item = nullptr; q->d_item_pos = nullptr;
    q->d_u = nullptr; q->d_v = nullptr; q->d_w = nullptr;
    q->item_cap = 0;
    const size_t cap = (size_t)(total + total / 8 + 1024);
    ArenaLayout a;
    const size_t o_t = a.add(sizeof(double) * cap), o_u = a.add(sizeof(double) * cap), o_v = a.add(sizeof(double) * cap),
                 o_w = a.add(sizeof(double) * cap), o_r = a.add(sizeof(int32_t) * cap), o_q = a.add(sizeof(int32_t) * cap),
                 o_s = a.add(sizeof(int32_t) * cap), o_p = a.add(sizeof(int32_t) * cap);
    char *base = nullptr;
    PMK_HIP(hipMalloc((void **)&base, a.bytes));
    q->d_item_t = reinterpret_cast<double *>(base + o_t);              // o_t == 0
    q->d_u = reinterpret_cast<double *>(base + o_u);
    q->d_v = reinterpret_cast<double *>(base + o_v);
    q->d_w = reinterpret_cast<double *>(base + o_w);
    q->d_item_region = reinterpret_cast<int32_t *>(base + o_r);
    q->d_item_query = reinterpret_cast<int32_t *>(base + o_q);
    q->d_sorted_item = reinterpret_cast<int32_t *>(base + o_s);
    q->d_item_pos = reinterpret_cast<int32_t *>(base + o_p);
    q->item_cap = (int64_t)cap;
    return 0;
}

}  // namespace pmk

extern "C" {

int pmk_query_create(pmk_model *m, int64_t Nq, const double *Xq, pmk_query **out)
{
    if (!out) { set_error("pmk_query_create: out is NULL"); return -4; }
    *out = nullptr;
    if (!m) { set_error("pmk_query_create: model is NULL"); return -1; }
    if (Nq < 0 || Nq > 0x7fffffff || (Nq > 0 && !Xq)) { set_error("pmk_query_create: bad Nq"); return -2; }
    if (m->P_global == 0) { set_error("pmk_query_create: attach a tree with pmk_model_set_bsp first"); return -1; }
    PMK_HIP(hipSetDevice(m->ctx->device));
    pmk_query *q = new (std::nothrow) pmk_query();
    if (!q) { set_error("out of memory"); return -100; }
    q->m = m; q->Nq = Nq; q->roff_P = m->P_global;
    int rc = query_reserve(q, std::max<int64_t>(Nq, 1));
    rc |= dev_alloc(&q->d_roff, m->P_global + 1);
    if (rc) { pmk_query_destroy(q); return -100; }
    if (Nq > 0) PMK_HIP(hipMemcpy(q->d_xq, Xq, sizeof(double) * (size_t)(Nq * m->D), hipMemcpyDefault));
    *out = q;
    return 0;
}

int pmk_query_create_items(pmk_model *m, int64_t n, const double *xq, const int32_t *region, pmk_query **out)
{
    if (!out) { set_error("pmk_query_create_items: out is NULL"); return -4; }
    *out = nullptr;
    if (n > 0 && (!region || !xq)) { set_error("pmk_query_create_items: NULL points or regions"); return -2; }
    pmk_query *q = nullptr;
    int rc = pmk_query_create(m, 0, nullptr, &q);
    if (rc) return rc;
    if ((rc = query_set_items(q, n, xq, region))) { pmk_query_destroy(q); return rc; }
    *out = q;
    return 0;
}

int pmk_query_export_requests(pmk_query *q, int64_t first, int64_t n, double *xq_dev, int32_t *region_dev)
{
    if (!q || !q->planned) { set_error("pmk_query_export_requests: query is not planned"); return -1; }
    if (first < 0 || n < 0 || first + n > q->total) { set_error("pmk_query_export_requests: bad item range"); return -3; }
    if (n > 0 && (!xq_dev || !region_dev)) { set_error("pmk_query_export_requests: NULL output"); return -2; }
    pmk_ctx *c = q->m->ctx;
    PMK_HIP(hipSetDevice(c->device));
    return launch_export_requests(q, first, n, xq_dev, region_dev, c->stream);
}

int pmk_query_export_results(pmk_query *q, double *u_dev, double *v_dev)
{
    if (!q || !q->planned) { set_error("pmk_query_export_results: query is not planned"); return -1; }
    pmk_ctx *c = q->m->ctx;
    PMK_HIP(hipSetDevice(c->device));
    return launch_export_results(q, u_dev, v_dev, c->stream);
}

int pmk_query_plan(pmk_query *q, double radius, double delta)
{
    if (!q) { set_error("pmk_query_plan: query is NULL"); return -1; }
    pmk_model *m = q->m;
    pmk_ctx *c = m->ctx;
    PMK_HIP(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    if (m->P_global != q->roff_P) {
        set_error("pmk_query_plan: the model's tree changed (%lld -> %lld leaves) after the query was created",
                  (long long)q->roff_P, (long long)m->P_global);
        return -4;
    }
    c->tic("plan");
    q->planned = false;
    q->total = 0;
    q->roff.assign((size_t)(m->P_global + 1), 0);
    if (q->Nq > 0) {
        int rc;
        PMK_HIP(hipMemsetAsync(q->d_cnt, 0, sizeof(int32_t) * (size_t)(q->Nq + 1), s));
        if ((rc = launch_plan_count(q, radius, delta, s))) return rc;
        if (exclusive_scan_i32_to_i64(q->d_cnt, q->d_qoff, q->Nq, &q->d_tmp, &q->tmp_bytes, s)) {
            set_error("pmk_query_plan: prefix scan failed");
            return -100;
        }
        PMK_HIP(hipMemcpyAsync(&q->total, q->d_qoff + q->Nq, sizeof(int64_t), hipMemcpyDeviceToHost, s));
        PMK_HIP(hipStreamSynchronize(s));
        if (q->total > 0x7fffffff) { set_error("pmk_query_plan: too many work items (%lld)", (long long)q->total); return -5; }
        if ((rc = grow_item_buffers(q, q->total))) return rc;
        if ((rc = launch_plan_fill(q, radius, delta, s))) return rc;
        if ((rc = launch_sort_items(q, s))) return rc;
        PMK_HIP(hipMemcpyAsync(q->roff.data(), q->d_roff, sizeof(int64_t) * q->roff.size(), hipMemcpyDeviceToHost, s));
        PMK_HIP(hipStreamSynchronize(s));
        if ((rc = PMK_BY_DTYPE(m, build_strip_tasks(q, s)))) return rc;
    }
    c->toc("plan");
    q->planned = true;
    return 0;
}

int pmk_query_counts(pmk_query *q, int64_t *total_items, int64_t *first_owned, int64_t *num_owned)
{
    if (!q || !q->planned) { set_error("pmk_query_counts: query is not planned"); return -1; }
    const pmk_model *m = q->m;
    if (total_items) *total_items = q->total;
    if (first_owned) *first_owned = q->roff[(size_t)m->leaf_base];
    if (num_owned) *num_owned = q->roff[(size_t)(m->leaf_base + m->P)] - q->roff[(size_t)m->leaf_base];
    return 0;
}

int pmk_query_region_offsets(pmk_query *q, int64_t *region_offsets)
{
    if (!q || !q->planned || !region_offsets) { set_error("pmk_query_region_offsets: query is not planned"); return -1; }
    std::memcpy(region_offsets, q->roff.data(), sizeof(int64_t) * q->roff.size());
    return 0;
}

int pmk_query_items(pmk_query *q, const pmk_kernel_desc *th)
{
    if (!q || !q->planned) { set_error("pmk_query_items: query is not planned"); return -1; }
    if (!kernel_ok(th)) { set_error("pmk_query_items: unknown kernel family"); return -2; }
    if (!q->m->fitted) { set_error("pmk_query_items: model is not fitted"); return -1; }
    pmk_ctx *c = q->m->ctx;
    PMK_HIP(hipSetDevice(c->device));
    c->tic("items");
    int rc = PMK_BY_DTYPE(q->m, launch_items(q, *th, c->stream));
    c->toc("items");
    return rc;
}

int pmk_query_item_buffers(pmk_query *q, void **u_dev, void **v_dev)
{
    if (!q || !q->planned) { set_error("pmk_query_item_buffers: query is not planned"); return -1; }
    if (u_dev) *u_dev = q->d_u;
    if (v_dev) *v_dev = q->d_v;
    return 0;
}

int pmk_query_mix(pmk_query *q, const pmk_kernel_desc *weight_th, int64_t q0, int64_t q1)
{
    if (!q || !q->planned) { set_error("pmk_query_mix: query is not planned"); return -1; }
    if (!kernel_ok(weight_th) || weight_th->family >= PMK_BB10) {
        set_error("pmk_query_mix: the blending profile must be a stationary kernel (evalkernel(tau, theta))");
        return -2;
    }
    if (q0 < 0 || q1 > q->Nq || q0 > q1) { set_error("pmk_query_mix: bad query range"); return -3; }
    pmk_ctx *c = q->m->ctx;
    PMK_HIP(hipSetDevice(c->device));
    c->tic("mix");
    int rc = launch_mix(q, *weight_th, q0, q1, c->stream);
    c->toc("mix");
    return rc;
}

int pmk_query_fetch(pmk_query *q, double *Yq, double *Vq)
{
    if (!q) { set_error("pmk_query_fetch: query is NULL"); return -1; }
    pmk_ctx *c = q->m->ctx;
    PMK_HIP(hipSetDevice(c->device));
    if (q->Nq > 0) {
        if (Yq) PMK_HIP(hipMemcpyAsync(Yq, q->d_yq, sizeof(double) * (size_t)q->Nq, hipMemcpyDeviceToHost, c->stream));
        if (Vq) PMK_HIP(hipMemcpyAsync(Vq, q->d_vq, sizeof(double) * (size_t)q->Nq, hipMemcpyDeviceToHost, c->stream));
    }
    PMK_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int pmk_query_debug(pmk_query *q, int64_t *home, int64_t *item_offsets, int64_t *item_region, double *item_t,
                    double *item_w, double *item_u, double *item_v)
{
    if (!q || !q->planned) { set_error("pmk_query_debug: query is not planned"); return -1; }
    pmk_ctx *c = q->m->ctx;
    PMK_HIP(hipSetDevice(c->device));
    PMK_HIP(hipStreamSynchronize(c->stream));
    const size_t Nq = (size_t)q->Nq, T = (size_t)q->total;
    if (home && Nq) {
        std::vector<int32_t> h(Nq);
        PMK_HIP(hipMemcpy(h.data(), q->d_home, sizeof(int32_t) * Nq, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < Nq; ++i) home[i] = h[i];
    }
    if (item_offsets) PMK_HIP(hipMemcpy(item_offsets, q->d_qoff, sizeof(int64_t) * (Nq + 1), hipMemcpyDeviceToHost));
    if (T == 0) return 0;
    if (item_region) {
        std::vector<int32_t> r(T);
        PMK_HIP(hipMemcpy(r.data(), q->d_item_region, sizeof(int32_t) * T, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < T; ++i) item_region[i] = r[i];
    }
    if (item_t) PMK_HIP(hipMemcpy(item_t, q->d_item_t, sizeof(double) * T, hipMemcpyDeviceToHost));
    if (item_w) PMK_HIP(hipMemcpy(item_w, q->d_w, sizeof(double) * T, hipMemcpyDeviceToHost));
    if (item_u || item_v) {
        std::vector<int32_t> pos(T);
        std::vector<double> tmp(T);
        PMK_HIP(hipMemcpy(pos.data(), q->d_item_pos, sizeof(int32_t) * T, hipMemcpyDeviceToHost));
        if (item_u) {
            PMK_HIP(hipMemcpy(tmp.data(), q->d_u, sizeof(double) * T, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < T; ++i) item_u[i] = tmp[(size_t)pos[i]];
        }
        if (item_v) {
            PMK_HIP(hipMemcpy(tmp.data(), q->d_v, sizeof(double) * T, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < T; ++i) item_v[i] = tmp[(size_t)pos[i]];
        }
    }
    return 0;
}

int pmk_predict_mixture(pmk_model *m, const pmk_kernel_desc *th, const pmk_kernel_desc *weight_th, int64_t Nq,
                        const double *Xq, double radius, double delta, double *Yq, double *Vq)
{
    if (!m) { set_error("pmk_predict_mixture: model is NULL"); return -1; }
    if (m->P_global != m->P || m->leaf_base != 0) {
        set_error("pmk_predict_mixture: the model holds %lld of %lld leaves; use the staged pmk_query_* calls",
                  (long long)m->P, (long long)m->P_global);
        return -1;
    }
    pmk_query *q = nullptr;
    int rc = pmk_query_create(m, Nq, Xq, &q);
    if (rc) return rc;
    if (!(rc = pmk_query_plan(q, radius, delta)) && !(rc = pmk_query_items(q, th)) &&
        !(rc = pmk_query_mix(q, weight_th, 0, Nq)))
        rc = pmk_query_fetch(q, Yq, Vq);
    pmk_query_destroy(q);
    return rc;
}

}  // extern "C"
