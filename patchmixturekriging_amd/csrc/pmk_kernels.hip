// K1 (kernel-matrix build), K5 (query-side BSP search + item list), K6 (mixture) and small helpers.

#include "pmk_device.h"

namespace pmk {

#define PMK_DISPATCH_D(D, CALL)                                    \
    switch (D) {                                                   \
    case 1: { constexpr int DD = 1; CALL; } break;                 \
    case 2: { constexpr int DD = 2; CALL; } break;                 \
    case 3: { constexpr int DD = 3; CALL; } break;                 \
    case 4: { constexpr int DD = 4; CALL; } break;                 \
    default: set_error("unsupported input dimension %d (1..%d)", (int)(D), MAX_D); return -2; \
    }

// Dense n x m kernel matrix for the host API (constructkernelmatrix, RKHS.jl:4-34 and :95-110).
// symmetric: entry (i,j) is evaluated as k(x_max, x_min) -- the lower-triangle value of the
// reference, mirrored -- so the result is exactly symmetric.
template <int D>
__global__ __launch_bounds__(256) void kmat_dense_kernel(pmk_kernel_desc th, int64_t n, const double *__restrict__ xs,
                                                         int64_t ldx, int64_t mcols, const double *__restrict__ zs,
                                                         int64_t ldz, double *__restrict__ K, int64_t ldk, int symmetric)
{
    const int64_t i = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const int64_t jb = (int64_t)blockIdx.y * 64 + (threadIdx.x >> 6) * 16;
    if (i >= n) return;
    double xi[D];
#pragma unroll
    for (int d = 0; d < D; ++d) xi[d] = xs[d * ldx + i];
    for (int64_t j = jb; j < jb + 16 && j < mcols; ++j) {
        double zj[D];
#pragma unroll
        for (int d = 0; d < D; ++d) zj[d] = zs[d * ldz + j];
        double v;
        if (symmetric && i < j) v = kern_eval<D>(th, zj, xi);
        else v = kern_eval<D>(th, xi, zj);
        K[i + j * ldk] = v;
    }
}

int launch_kernel_matrix_dense(const pmk_kernel_desc &th, int D, int64_t n, const double *d_xs, int64_t ldx,
                               int64_t mcols, const double *d_zs, int64_t ldz, double *d_K, int64_t ldk,
                               bool symmetric, hipStream_t s)
{
    dim3 grid((unsigned)((n + 63) / 64), (unsigned)((mcols + 63) / 64));
    PMK_DISPATCH_D(D, hipLaunchKernelGGL(kmat_dense_kernel<DD>, grid, dim3(256), 0, s, th, n, d_xs, ldx, mcols, d_zs,
                                         ldz, d_K, ldk, symmetric ? 1 : 0));
    PMK_HIP(hipGetLastError());
    return 0;
}

// query! of the single-problem path (src/RKHS/RKHS.jl:220-247): Yq[j] = sum_i k(xq_j, x_i) c_i; with MULTI the kernel
// of centre i is ths[i] (the method for RKHSProblemType{Vector{KT}}, RKHS.jl:278-305)
template <int D, bool MULTI>
__global__ __launch_bounds__(64) void query_mean_kernel(pmk_kernel_desc th, const pmk_kernel_desc *__restrict__ ths, int64_t n,
                                                        const double *__restrict__ xs, int64_t ldx,
                                                        const double *__restrict__ c, int64_t nq,
                                                        const double *__restrict__ xq, double *__restrict__ yq)
{
    const int64_t j = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (j >= nq) return;
    double q[D];
#pragma unroll
    for (int d = 0; d < D; ++d) q[d] = xq[j * D + d];
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        double xi[D];
#pragma unroll
        for (int d = 0; d < D; ++d) xi[d] = xs[d * ldx + i];
        s += kern_eval<D>(MULTI ? ths[i] : th, q, xi) * c[i];
    }
    yq[j] = s;
}

// th: ONE kernel (host pointer, nth == 1) or n kernels, one per centre (DEVICE pointer, nth == n)
int launch_query_mean(const pmk_kernel_desc *th, int nth, int D, int64_t n, const double *d_xs, int64_t ldx, const double *d_c,
                      int64_t nq, const double *d_xq, double *d_yq, hipStream_t s)
{
    dim3 grid((unsigned)((nq + 63) / 64));
    if (nth == 1) {
        PMK_DISPATCH_D(D, hipLaunchKernelGGL((query_mean_kernel<DD, false>), grid, dim3(64), 0, s, *th, nullptr, n, d_xs, ldx,
                                             d_c, nq, d_xq, d_yq));
    } else {
        PMK_DISPATCH_D(D, hipLaunchKernelGGL((query_mean_kernel<DD, true>), grid, dim3(64), 0, s, pmk_kernel_desc{}, th, n, d_xs,
                                             ldx, d_c, nq, d_xq, d_yq));
    }
    PMK_HIP(hipGetLastError());
    return 0;
}

// =============================================================================================
// K5: home leaf + neighbour items per query.  Bit-exact restatement of findpartition
// (src/patchwork/partition.jl:248-262) and findneighbourpartitions (src/RKHS/mixtureGP.jl:339-405):
// no FMA contraction, the reference's operation order.  One thread per query; every hyperplane of
// the tree is visited in pre-order (the reference does not prune either).
// =============================================================================================
#pragma clang fp contract(off)
template <int D>
__device__ __forceinline__ int find_leaf(const double *__restrict__ hv, const double *__restrict__ hc, int levels,
                                         int64_t P, const double *x, int dot_mode)
{
    int node = 0;
    for (int l = 1; l < levels; ++l) {
        double u[D];
#pragma unroll
        for (int d = 0; d < D; ++d) u[d] = hv[node * D + d];
        node = (dot_seq<D>(u, x, dot_mode) < hc[node]) ? 2 * node + 1 : 2 * node + 2;
    }
    return node - (int)(P - 1);
}

// The tree (heap-order normals and offsets + the pre-order permutation) is staged in LDS once per
// workgroup when it fits (P <= PLAN_LDS_NODES + 1): the hyperplane loop then reads LDS broadcasts instead
// of issuing a dependent global load per hyperplane and per tree level.
constexpr int PLAN_LDS_NODES = 2047;
// The count pass keeps the first PLAN_STAGE neighbour hits of every query (region, t) in a staging array; the fill pass
// copies them instead of walking the hyperplanes a second time (a workgroup in which some query has more hits than
// that walks again: same values either way).  Config C: 0.44 neighbours per query on average, the fill pass drops from
// 1.18 ms to a copy.
constexpr int PLAN_STAGE = 4;

template <int D, bool FILL, bool LDS>
__global__ __launch_bounds__(256) void plan_kernel(int64_t Nq, const double *__restrict__ xq,
                                                   const double *__restrict__ hv_g, const double *__restrict__ hc_g,
                                                   const int32_t *__restrict__ pre_g, int levels, int dot_mode, int64_t P,
                                                   double radius, double delta, int32_t *__restrict__ home_out,
                                                   int32_t *__restrict__ cnt_out, const int64_t *__restrict__ qoff,
                                                   int32_t *__restrict__ item_region, double *__restrict__ item_t,
                                                   int32_t *__restrict__ item_query, int32_t *__restrict__ stage_r,
                                                   double *__restrict__ stage_t, int64_t stage_ld)
{
    extern __shared__ double plan_sm[];
    const double *hv = hv_g, *hc = hc_g;
    const int32_t *pre = pre_g;
    if (FILL) {
        const int64_t jq = (int64_t)blockIdx.x * 256 + threadIdx.x;
        const int cn = jq < Nq ? (int)(qoff[jq + 1] - qoff[jq]) - 1 : 0;          // neighbour hits of this query
        if (!__syncthreads_or(cn > PLAN_STAGE)) {
            if (jq >= Nq) return;
            const int64_t b0 = qoff[jq];
            for (int e = 0; e < cn; ++e) {
                item_region[b0 + e] = stage_r[e * stage_ld + jq];
                item_t[b0 + e] = stage_t[e * stage_ld + jq];
                item_query[b0 + e] = (int32_t)jq;
            }
            item_region[b0 + cn] = home_out[jq];
            item_t[b0 + cn] = 0.0;
            item_query[b0 + cn] = (int32_t)jq;
            return;
        }
    }
    int tree_doubles = 0;
    if (LDS) {
        const int nn = (int)(P - 1);
        double *sv = plan_sm, *sc = plan_sm + (size_t)nn * D;
        int32_t *sp = reinterpret_cast<int32_t *>(sc + nn);
        for (int e = threadIdx.x; e < nn * D; e += 256) sv[e] = hv_g[e];
        for (int e = threadIdx.x; e < nn; e += 256) { sc[e] = hc_g[e]; sp[e] = pre_g[e]; }
        hv = sv; hc = sc; pre = sp;
        tree_doubles = nn * (D + 1) + (nn + 1) / 2;
    }
    // Per-wave scratch behind the tree.  A query is within `radius` of a given (infinite) hyperplane with a few per cent
    // probability, whatever the level of the plane -- so with 64 unrelated queries in a wave nearly EVERY hyperplane has a
    // candidate in some lane, and the two leaf searches behind the distance test (2 x levels dependent loads) ran for the
    // whole wave almost every time: 25 x the work.  Candidates (lane, plane) are therefore queued per wave and evaluated 64
    // at a time, one per lane, on the source lane's point (same operations in the same order: same bits); an accepted
    // candidate sets a bit in the source lane's mask, and each lane then walks its own few bits in pre-order to emit its
    // items exactly as the direct loop did.  Planes are taken in chunks of 256 (8 mask words per lane).
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int WAVE_DOUBLES = 64 * D + 32 + 256 + 64;      // points, homes (int32), masks (8 x uint32), queue (128 x uint32)
    double *wsm = plan_sm + tree_doubles + wave * WAVE_DOUBLES;
    double *qp = wsm;
    int32_t *qhome = reinterpret_cast<int32_t *>(wsm + 64 * D);
    uint32_t *bits = reinterpret_cast<uint32_t *>(wsm + 64 * D + 32);
    uint32_t *queue = reinterpret_cast<uint32_t *>(wsm + 64 * D + 32 + 256);
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = j < Nq;
    double p[D];
#pragma unroll
    for (int d = 0; d < D; ++d) p[d] = valid ? xq[j * D + d] : 0.0;
    __syncthreads();                                          // the tree is staged
    const int home = find_leaf<D>(hv, hc, levels, P, p, dot_mode);
#pragma unroll
    for (int d = 0; d < D; ++d) qp[lane * D + d] = p[d];
    qhome[lane] = home;
    int count = 0;
    int64_t base = 0;
    if (FILL && valid) base = qoff[j];
    double r2lo = -1.0, r2hi = __builtin_inf();                     // radius^2 not representable well: always take the root
    if (radius > 1e-140 && radius < 1e150) {
        // sqrt(s) < radius decided without the root wherever s is clear of radius^2 (sqrt is correctly rounded and
        // monotone); the root itself is taken only inside the band: same decisions
        const double r2 = radius * radius;
        r2lo = r2 * (1.0 - 0x1p-50);
        r2hi = r2 * (1.0 + 0x1p-50);
    } else if (!(radius > 0.0)) {
        r2hi = -1.0;                                                 // radius <= 0 or NaN: sqrt(s) < radius never holds
    }
    auto wave_sync = [] {                                            // LDS traffic of ONE wave: in order, nothing to wait for
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");       // but the compiler must not move it
        __builtin_amdgcn_wave_barrier();
    };
    // distance of point x to hyperplane h along its normal; true when within radius (mixtureGP.jl:361-367)
    auto plane_t = [&](int h, const double *x, double *u, double &tt) -> bool {
#pragma unroll
        for (int d = 0; d < D; ++d) u[d] = hv[h * D + d];
        const double c = hc[h];
        tt = -dot_seq<D>(u, x, dot_mode) + c;                        // :361
        double r0 = (x[0] + tt * u[0]) - x[0];                       // z = p + t.*u ; norm(z - p)   :362,:367
        double s = r0 * r0;
#pragma unroll
        for (int d = 1; d < D; ++d) {
            double r = (x[d] + tt * u[d]) - x[d];
            s = s + r * r;
        }
        return s < r2lo || (s <= r2hi && sqrt(s) < radius);
    };
    // the two leaf searches behind an accepted distance test: the region across the plane, or -1 (:374-388)
    auto across = [&](const double *x, const double *u, double tt, int hm) -> int {
        const double tp = tt + delta, tm = tt - delta;
        double z1[D], z2[D];
#pragma unroll
        for (int d = 0; d < D; ++d) { z1[d] = x[d] + tp * u[d]; z2[d] = x[d] + tm * u[d]; }
        const int r1 = find_leaf<D>(hv, hc, levels, P, z1, dot_mode);
        const int r2 = find_leaf<D>(hv, hc, levels, P, z2, dot_mode);
        if ((r2 == hm) != (r1 == hm)) return (r1 == hm) ? r2 : r1;   // xor :388
        return -1;
    };
    const int nn = (int)(P - 1);
    for (int c0 = 0; c0 < nn; c0 += 256) {
        const int c1 = min(nn, c0 + 256);
#pragma unroll
        for (int w = 0; w < 8; ++w) bits[lane * 8 + w] = 0u;
        int qn = 0;                                                  // queued candidates (the same in every lane)
        auto run_batch = [&](int n) {                                // entries 0..n-1 of the queue, one per lane
            wave_sync();
            if (lane < n) {
                const uint32_t ent = queue[lane];
                const int src = (int)(ent & 255u), il = (int)(ent >> 8);
                double xs[D], u[D], tt;
#pragma unroll
                for (int d = 0; d < D; ++d) xs[d] = qp[src * D + d];
                (void)plane_t(pre[c0 + il], xs, u, tt);
                if (across(xs, u, tt, qhome[src]) >= 0) atomicOr(&bits[src * 8 + (il >> 5)], 1u << (il & 31));
            }
            wave_sync();
        };
        for (int i = c0; i < c1; ++i) {
            double u[D], tt;
            const bool cand = plane_t(pre[i], p, u, tt) && valid;
            const unsigned long long m = __ballot(cand);
            if (m == 0ull) continue;
            const int before = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (cand) queue[qn + before] = (uint32_t)lane | ((uint32_t)(i - c0) << 8);
            qn += __popcll(m);
            if (qn >= 64) {
                run_batch(64);
                const uint32_t carry = lane < qn - 64 ? queue[64 + lane] : 0u;
                wave_sync();
                if (lane < qn - 64) queue[lane] = carry;
                qn -= 64;
            }
        }
        if (qn > 0) run_batch(qn);
        // emit this lane's accepted planes in pre-order
        for (int w = 0; w < 8; ++w) {
            uint32_t mk = bits[lane * 8 + w];
            while (mk) {
                const int bpos = __builtin_ctz(mk);
                mk &= mk - 1u;
                double u[D], tt;
                (void)plane_t(pre[c0 + 32 * w + bpos], p, u, tt);
                const int reg = across(p, u, tt, home);
                if (FILL) {
                    item_region[base + count] = reg;
                    item_t[base + count] = tt;
                    item_query[base + count] = (int32_t)j;
                } else if (count < PLAN_STAGE) {
                    stage_r[count * stage_ld + j] = reg;
                    stage_t[count * stage_ld + j] = tt;
                }
                ++count;
            }
        }
        wave_sync();
    }
    if (!valid) return;
    if (FILL) {
        item_region[base + count] = home;     // home region last (mixtureGP.jl:237-239)
        item_t[base + count] = 0.0;
        item_query[base + count] = (int32_t)j;
    } else {
        home_out[j] = home;
        cnt_out[j] = count + 1;
    }
}
#pragma clang fp contract(fast)

template <int D>
static int launch_plan_D(pmk_query *q, double radius, double delta, bool fill, hipStream_t s)
{
    const pmk_model *m = q->m;
    dim3 grid((unsigned)((q->Nq + 255) / 256));
    const int64_t nn = m->P_global - 1;
    const bool lds = nn <= PLAN_LDS_NODES;
    // the tree (when it fits) + the four waves' scratch (plan_kernel: WAVE_DOUBLES)
    const size_t tree_doubles = lds ? (size_t)nn * (D + 1) + (size_t)(nn + 1) / 2 : 0;
    const size_t bytes = sizeof(double) * (tree_doubles + 4 * (size_t)(64 * D + 32 + 256 + 64));
#define PMK_PLAN(FILL_, LDS_)                                                                                          \
    hipLaunchKernelGGL((plan_kernel<D, FILL_, LDS_>), grid, dim3(256), bytes, s, q->Nq, q->d_xq, m->d_hv, m->d_hc,      \
                       m->d_pre, m->levels, m->dot_mode, m->P_global, radius, delta, q->d_home, q->d_cnt, q->d_qoff,                \
                       q->d_item_region, q->d_item_t, q->d_item_query, q->d_stage_r, q->d_stage_t, q->nq_cap)
    if (fill) { if (lds) PMK_PLAN(true, true); else PMK_PLAN(true, false); }
    else      { if (lds) PMK_PLAN(false, true); else PMK_PLAN(false, false); }
#undef PMK_PLAN
    PMK_HIP(hipGetLastError());
    return 0;
}

// the LDS-staged tree of plan_kernel needs up to nn (D + 1) 8 + 4 nn bytes = 90 KB at nn = 2047, D = 4: above the
// default 64 KB dynamic limit.  Per device, called by pmk_ctx_create with the context's device current.
template <int D>
static int set_plan_attributes_D()
{
    constexpr int bytes = 8 * (PLAN_LDS_NODES * (D + 1) + (PLAN_LDS_NODES + 1) / 2 + 4 * (64 * D + 32 + 256 + 64));
    PMK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(plan_kernel<D, true, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    PMK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(plan_kernel<D, false, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    return 0;
}

int set_plan_attributes()
{
    int rc = set_plan_attributes_D<1>();
    if (!rc) rc = set_plan_attributes_D<2>();
    if (!rc) rc = set_plan_attributes_D<3>();
    if (!rc) rc = set_plan_attributes_D<4>();
    return rc;
}

static int launch_plan(pmk_query *q, double radius, double delta, bool fill, hipStream_t s)
{
    int rc = 0;
    PMK_DISPATCH_D(q->m->D, rc = launch_plan_D<DD>(q, radius, delta, fill, s));
    return rc;
}

int launch_plan_count(pmk_query *q, double radius, double delta, hipStream_t s) { return launch_plan(q, radius, delta, false, s); }
int launch_plan_fill(pmk_query *q, double radius, double delta, hipStream_t s) { return launch_plan(q, radius, delta, true, s); }

// ---------------------------------------------------------------------------------------------
// Exclusive prefix sums of the per-query item counts: d_out[i] = sum of d_in[0..i), i = 0..n (n + 1 outputs; d_in needs
// n + 1 readable entries, the last one is ignored).  Three small launches: sums of 2048-entry blocks, their scan by one
// workgroup, the blocks again with their offsets.
// ---------------------------------------------------------------------------------------------
constexpr int SCAN_PER_THREAD = 8, SCAN_BLOCK = 256 * SCAN_PER_THREAD;

// sum over the workgroup of one value per thread (256 threads); result in every thread
__device__ __forceinline__ int64_t block_sum_256(int64_t v, int64_t *red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void scan_block_sums_kernel(const int32_t *__restrict__ in, int64_t n, int64_t *__restrict__ bsum)
{
    __shared__ int64_t red[4];
    const int64_t i0 = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_PER_THREAD;
    int64_t v = 0;
#pragma unroll
    for (int u = 0; u < SCAN_PER_THREAD; ++u)
        if (i0 + u < n) v += in[i0 + u];
    v = block_sum_256(v, red);
    if (threadIdx.x == 0) bsum[blockIdx.x] = v;
}

// in place: bsum[b] <- sum of bsum[0..b); one workgroup, 256 entries at a time
__global__ __launch_bounds__(256) void scan_offsets_kernel(int64_t *__restrict__ bsum, int64_t nb)
{
    __shared__ int64_t sh[256];
    int64_t carry = 0;
    for (int64_t b0 = 0; b0 < nb; b0 += 256) {
        const int64_t b = b0 + threadIdx.x;
        const int64_t mine = b < nb ? bsum[b] : 0;
        sh[threadIdx.x] = mine;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {                  // inclusive scan (Hillis-Steele)
            const int64_t add = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += add;
            __syncthreads();
        }
        if (b < nb) bsum[b] = carry + sh[threadIdx.x] - mine;
        carry += sh[255];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void scan_apply_kernel(const int32_t *__restrict__ in, int64_t n, const int64_t *__restrict__ boff,
                                                         int64_t *__restrict__ out)
{
    __shared__ int64_t sh[256];
    const int64_t i0 = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_PER_THREAD;
    int32_t v[SCAN_PER_THREAD];
    int64_t mine = 0;
#pragma unroll
    for (int u = 0; u < SCAN_PER_THREAD; ++u) {
        v[u] = i0 + u < n ? in[i0 + u] : 0;
        mine += v[u];
    }
    sh[threadIdx.x] = mine;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const int64_t add = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
    }
    int64_t run = boff[blockIdx.x] + sh[threadIdx.x] - mine;
#pragma unroll
    for (int u = 0; u < SCAN_PER_THREAD; ++u) {
        if (i0 + u <= n) out[i0 + u] = run;                  // n + 1 outputs
        run += v[u];
    }
}

int64_t exclusive_scan_i32_to_i64(const int32_t *d_in, int64_t *d_out, int64_t n, void **tmp, size_t *tmp_bytes,
                                  hipStream_t s)
{
    const int64_t nb = (n + 1 + SCAN_BLOCK - 1) / SCAN_BLOCK;          // blocks over the n + 1 outputs
    const size_t need = sizeof(int64_t) * (size_t)nb;
    if (need > *tmp_bytes) {
        if (*tmp) (void)hipFree(*tmp);
        *tmp = nullptr; *tmp_bytes = 0;
        if (hipMalloc(tmp, need) != hipSuccess) return -1;
        *tmp_bytes = need;
    }
    int64_t *bsum = reinterpret_cast<int64_t *>(*tmp);
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3((unsigned)nb), dim3(256), 0, s, d_in, n, bsum);
    hipLaunchKernelGGL(scan_offsets_kernel, dim3(1), dim3(256), 0, s, bsum, nb);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nb), dim3(256), 0, s, d_in, n, (const int64_t *)bsum, d_out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

__global__ void iota_kernel(int32_t *v, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) v[i] = (int32_t)i;
}

int launch_iota(int32_t *d, int64_t n, hipStream_t s)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d, n);
    PMK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Stable counting sort of the items by region (keys < P, a few thousand at most): sorted_item[pos] = item,
// item_pos[item] = pos, roff[r] = first position of region r (roff[P] = n).  Deterministic, so every rank of a multi-GPU
// job derives the same order from a replicated plan.  Items are cut into blocks of `bitems` consecutive items:
//   sort_hist_kernel     hist[r][b]  = items of region r in block b
//   sort_scan_kernel     hist[r][b] <- items of region r in blocks < b ; total[r]    (one wave per region)
//   sort_roff_kernel     roff[r]     = items of regions < r                          (one workgroup)
//   sort_scatter_kernel  one wave per block walks its items 64 at a time in order: the lanes that hold the same region
//                        form a group (found with ballots over the distinct regions of the tile), a lane's rank in its
//                        group is its position in it, the group's first lane takes the group's base with ONE atomic add
//                        on hist[r][b] -- all groups of a tile at once -- and hands it to the others
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sort_hist_kernel(const int32_t *__restrict__ key, int64_t n, int bitems, int64_t nb,
                                                        int64_t P, int32_t *__restrict__ hist)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int k = key[i];
    if ((unsigned)k >= (unsigned)P) return;                 // an invalid region of an explicit item list (reported by its check)
    atomicAdd(&hist[(int64_t)k * nb + i / bitems], 1);
}

// the same with the block's histogram in LDS (P <= SORT_LDS_BINS): one workgroup per block of items, no memset before
constexpr int SORT_LDS_BINS = 8192;
__global__ __launch_bounds__(256) void sort_hist_lds_kernel(const int32_t *__restrict__ key, int64_t n, int bitems, int64_t nb,
                                                            int64_t P, int32_t *__restrict__ hist)
{
    extern __shared__ int32_t sort_bins[];
    for (int r = threadIdx.x; r < (int)P; r += 256) sort_bins[r] = 0;
    __syncthreads();
    const int64_t b = blockIdx.x, i0 = b * bitems, i1 = min(n, i0 + (int64_t)bitems);
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
        const int k = key[i];
        if ((unsigned)k < (unsigned)P) atomicAdd(&sort_bins[k], 1);
    }
    __syncthreads();
    for (int r = threadIdx.x; r < (int)P; r += 256) hist[(int64_t)r * nb + b] = sort_bins[r];
}

__global__ __launch_bounds__(256) void sort_scan_kernel(int32_t *__restrict__ hist, int64_t nb, int64_t P, int32_t *__restrict__ total)
{
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= P) return;
    int32_t *h = hist + r * nb;
    int32_t carry = 0;
    for (int64_t b0 = 0; b0 < nb; b0 += 64) {
        const int64_t b = b0 + lane;
        const int32_t mine = b < nb ? h[b] : 0;
        int32_t inc = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int32_t up = __shfl_up(inc, o);
            if (lane >= o) inc += up;
        }
        if (b < nb) h[b] = carry + inc - mine;
        carry += __shfl(inc, 63);
    }
    if (lane == 0) total[r] = carry;
}

__global__ __launch_bounds__(256) void sort_roff_kernel(const int32_t *__restrict__ total, int64_t P, int64_t *__restrict__ roff)
{
    __shared__ int64_t sh[256];
    int64_t carry = 0;
    for (int64_t r0 = 0; r0 < P; r0 += 256) {
        const int64_t r = r0 + threadIdx.x;
        const int64_t mine = r < P ? total[r] : 0;
        sh[threadIdx.x] = mine;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const int64_t add = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += add;
            __syncthreads();
        }
        if (r < P) roff[r] = carry + sh[threadIdx.x] - mine;
        carry += sh[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) roff[P] = carry;
}

__global__ __launch_bounds__(64) void sort_scatter_kernel(const int32_t *__restrict__ key, int64_t n, int bitems, int64_t nb,
                                                          int64_t P, int32_t *__restrict__ hist, const int64_t *__restrict__ roff,
                                                          int32_t *__restrict__ sorted_item, int32_t *__restrict__ item_pos)
{
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x, i0 = b * bitems, i1 = min(n, i0 + (int64_t)bitems);
    // one tile behind: the atomic adds of tile t are in flight while the groups of tile t + 1 are formed
    int64_t pi = 0, proff = 0;
    int32_t pbase = 0;
    int prank = 0, pleader = 0;
    bool pvalid = false;
    for (int64_t t0 = i0; t0 < i1 + 64; t0 += 64) {
        const int64_t i = t0 + lane;
        const int k = i < i1 ? key[i] : -1;
        const bool valid = (unsigned)k < (unsigned)P;       // invalid regions are left out (see sort_hist_kernel)
        // groups of equal keys: rank of the lane in its group, the group's size and first lane
        int rank = 0, cnt = 0, leader = lane;
        unsigned long long todo = __ballot(valid);
        while (todo) {
            const int first = __builtin_ctzll(todo);
            const int kf = __shfl(k, first);
            const unsigned long long grp = __ballot(valid && k == kf);
            if (k == kf && valid) {
                rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(grp >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)grp, 0u));
                cnt = __popcll(grp);
                leader = first;
            }
            todo &= ~grp;
        }
        const int64_t ro = valid ? roff[k] : 0;
        // finish the previous tile: its atomic adds have had the time of the loop above to return
        const int32_t pb = __shfl(pbase, pleader);
        if (pvalid) {
            const int64_t pos = proff + pb + prank;
            sorted_item[pos] = (int32_t)pi;
            item_pos[pi] = (int32_t)pos;
        }
        // this tile's groups take their bases only now, behind the previous tile's returned values: a later tile gets the
        // later base
        int32_t base = 0;
        if (valid && lane == leader) base = atomicAdd(&hist[(int64_t)k * nb + b], cnt);     // items of this region placed so far
        pi = i; proff = ro; pbase = base; prank = rank; pleader = leader; pvalid = valid;
    }
}

int launch_sort_items(pmk_query *q, hipStream_t s)
{
    const int64_t n = q->total;
    const pmk_model *m = q->m;
    if (n == 0) return 0;
    const int64_t P = m->P_global;
    // block length: at least two items per region and block on average, so that the histogram stays smaller than the items
    const int bitems = (int)std::max<int64_t>(1024, 64 * ((2 * P + 63) / 64));
    const int64_t nb = (n + bitems - 1) / bitems;
    const int64_t words = P * nb + P + 64;                   // hist[P][nb], total[P]
    if (q->sort_cap < words) {
        if (q->d_sort_scratch) PMK_HIP(hipFree(q->d_sort_scratch));
        q->d_sort_scratch = nullptr;
        q->sort_cap = 0;
        PMK_HIP(hipMalloc(&q->d_sort_scratch, sizeof(int32_t) * (size_t)(words + words / 8)));
        q->sort_cap = words + words / 8;
    }
    int32_t *hist = reinterpret_cast<int32_t *>(q->d_sort_scratch);
    int32_t *total = hist + P * nb;
    if (P <= SORT_LDS_BINS) {
        hipLaunchKernelGGL(sort_hist_lds_kernel, dim3((unsigned)nb), dim3(256), sizeof(int32_t) * (size_t)P, s, q->d_item_region, n,
                           bitems, nb, P, hist);
    } else {
        PMK_HIP(hipMemsetAsync(hist, 0, sizeof(int32_t) * (size_t)(P * nb), s));
        hipLaunchKernelGGL(sort_hist_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, q->d_item_region, n, bitems, nb, P, hist);
    }
    hipLaunchKernelGGL(sort_scan_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, s, hist, nb, P, total);
    hipLaunchKernelGGL(sort_roff_kernel, dim3(1), dim3(256), 0, s, total, P, q->d_roff);
    hipLaunchKernelGGL(sort_scatter_kernel, dim3((unsigned)nb), dim3(64), 0, s, q->d_item_region, n, bitems, nb, P, hist,
                       (const int64_t *)q->d_roff, q->d_sorted_item, q->d_item_pos);
    PMK_HIP(hipGetLastError());
    return 0;
}

// ---- multi-GPU request/response helpers (query-sharded predict) ----
// requests of the sorted items [first, first + n): coordinates (point-major) and global region
template <int D>
__global__ void export_requests_kernel(int64_t first, int64_t n, const int32_t *__restrict__ sorted_item,
                                       const int32_t *__restrict__ item_query, const int32_t *__restrict__ item_region,
                                       const double *__restrict__ xq, double *__restrict__ x_out,
                                       int32_t *__restrict__ region_out)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const int32_t it = sorted_item[first + k];
    const int64_t j = item_query[it];
#pragma unroll
    for (int d = 0; d < D; ++d) x_out[k * D + d] = xq[j * D + d];
    region_out[k] = item_region[it];
}

int launch_export_requests(pmk_query *q, int64_t first, int64_t n, double *x_out, int32_t *region_out, hipStream_t s)
{
    if (n == 0) return 0;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    switch (q->m->D) {
    case 1: hipLaunchKernelGGL(export_requests_kernel<1>, grid, block, 0, s, first, n, q->d_sorted_item, q->d_item_query, q->d_item_region, q->d_xq, x_out, region_out); break;
    case 2: hipLaunchKernelGGL(export_requests_kernel<2>, grid, block, 0, s, first, n, q->d_sorted_item, q->d_item_query, q->d_item_region, q->d_xq, x_out, region_out); break;
    case 3: hipLaunchKernelGGL(export_requests_kernel<3>, grid, block, 0, s, first, n, q->d_sorted_item, q->d_item_query, q->d_item_region, q->d_xq, x_out, region_out); break;
    case 4: hipLaunchKernelGGL(export_requests_kernel<4>, grid, block, 0, s, first, n, q->d_sorted_item, q->d_item_query, q->d_item_region, q->d_xq, x_out, region_out); break;
    default: set_error("input dimension %d not supported", q->m->D); return -2;
    }
    PMK_HIP(hipGetLastError());
    return 0;
}

// (u, v) back into item (reference) order
__global__ void export_results_kernel(int64_t n, const int32_t *__restrict__ item_pos, const double *__restrict__ u,
                                      const double *__restrict__ v, double *__restrict__ u_out, double *__restrict__ v_out)
{
    const int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (it >= n) return;
    const int32_t pos = item_pos[it];
    if (u_out) u_out[it] = u[pos];
    if (v_out) v_out[it] = v[pos];
}

int launch_export_results(pmk_query *q, double *u_out, double *v_out, hipStream_t s)
{
    if (q->total == 0) return 0;
    hipLaunchKernelGGL(export_results_kernel, dim3((unsigned)((q->total + 255) / 256)), dim3(256), 0, s, q->total,
                       q->d_item_pos, q->d_u, q->d_v, u_out, v_out);
    PMK_HIP(hipGetLastError());
    return 0;
}

// explicit items: one item per point; flag regions outside [lo, hi)
__global__ void explicit_items_kernel(int64_t n, const int32_t *__restrict__ region, int32_t lo, int32_t hi,
                                      int32_t *__restrict__ item_query, double *__restrict__ item_t,
                                      int64_t *__restrict__ qoff, int32_t *__restrict__ home, int *__restrict__ bad)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k > n) return;
    qoff[k] = k;
    if (k == n) return;
    item_query[k] = (int32_t)k;
    item_t[k] = 0.0;
    const int32_t r = region[k];
    home[k] = r;
    if (r < lo || r >= hi) atomicOr(bad, 1);
}

int launch_explicit_items(pmk_query *q, int *d_bad, hipStream_t s)
{
    const pmk_model *m = q->m;
    hipLaunchKernelGGL(explicit_items_kernel, dim3((unsigned)((q->Nq + 256) / 256)), dim3(256), 0, s, q->Nq,
                       q->d_item_region, (int32_t)m->leaf_base, (int32_t)(m->leaf_base + m->P), q->d_item_query,
                       q->d_item_t, q->d_qoff, q->d_home, d_bad);
    PMK_HIP(hipGetLastError());
    return 0;
}

// =============================================================================================
// K6: mixture weights and blend, src/RKHS/mixtureGP.jl:224-272.  Neighbour weights phi_w(|t|) in
// hyperplane order, home weight 1 last, normalise, Yq = sum w u, Vq = sum w (v w).
// =============================================================================================
__global__ __launch_bounds__(256) void mix_kernel(int64_t q0, int64_t q1, const int64_t *__restrict__ qoff,
                                                  const double *__restrict__ item_t, const int32_t *__restrict__ item_pos,
                                                  const double *__restrict__ u, const double *__restrict__ v,
                                                  pmk_kernel_desc wth, double *__restrict__ w_out,
                                                  double *__restrict__ yq, double *__restrict__ vq)
{
    const int64_t j = q0 + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= q1) return;
    const int64_t b = qoff[j], e = qoff[j + 1];
    double sw = 0.0;
    for (int64_t it = b; it < e; ++it) {
        const double w = (it == e - 1) ? 1.0 : profile(wth, fabs(item_t[it]));
        w_out[it] = w;
        sw = (it == b) ? w : sw + w;
    }
    double y = 0.0, vv = 0.0;
    for (int64_t it = b; it < e; ++it) {
        const double w = w_out[it] / sw;
        const int32_t pos = item_pos[it];
        const double ui = u[pos], vi = v[pos];
        y = (it == b) ? w * ui : y + w * ui;
        vv = (it == b) ? w * (vi * w) : vv + w * (vi * w);
    }
    yq[j] = y;
    vq[j] = vv;
}

int launch_mix(pmk_query *q, const pmk_kernel_desc &wth, int64_t q0, int64_t q1, hipStream_t s)
{
    if (q1 <= q0) return 0;
    hipLaunchKernelGGL(mix_kernel, dim3((unsigned)((q1 - q0 + 255) / 256)), dim3(256), 0, s, q0, q1, q->d_qoff,
                       q->d_item_t, q->d_item_pos, q->d_u, q->d_v, wth, q->d_w, q->d_yq, q->d_vq);
    PMK_HIP(hipGetLastError());
    return 0;
}

}  // namespace pmk
