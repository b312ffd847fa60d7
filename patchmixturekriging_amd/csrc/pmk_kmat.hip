// K1: batched kernel-matrix build into the factorisation slabs (compiled for real = double and float).
#include "pmk_real.h"

namespace pmk {
namespace PMK_NS {

// =============================================================================================
// K1: batched kernel-matrix build into the factorisation slabs.
// Replaces the evalkernel real loop of constructkernelmatrix! (src/RKHS/RKHS.jl:21-31) and the
// diagonal "+= sigma2" of fitmixtureGP! (src/RKHS/mixtureGP.jl:102-104).  HBM-write bound:
// 8 * ld^2 / 2 bytes per patch (lower triangle; diagonal 64x64 tiles are written whole and exactly
// symmetric).  One workgroup = one 64 x 64 tile, one thread = 4 contiguous rows x 4 columns, so a
// 16-thread row group stores 512 contiguous bytes per column.  Padding rows/columns (index >= n) are
// written as identity so the padded factorisation stays positive definite.
// =============================================================================================
// stage < 0: the whole lower triangle of every patch (grid.x = its 64 x 64 tiles).  stage >= 0: ONE 128-wide block
// column per patch -- the one the factorisation needs next: block column 0 of every patch at stage 0, block column
// c >= 1 of patch p at stage c + (max_nt - nt_p) (the end-aligned schedule of the step launches, which read block
// columns <= k + 1 at launch l = k + max_nt - nt_p, i.e. stages <= l + 1); grid.x = 2 * (64-row tiles of the tallest patch).
template <int D, int FAM>
__global__ __launch_bounds__(256) void kmat_slab_kernel(const PatchDesc *__restrict__ descs, const real *__restrict__ x,
                                                        real *__restrict__ A, pmk_kernel_desc th, double sigma2_d, int stage,
                                                        int max_nt, const real *__restrict__ dg)
{
    const PatchDesc pd = descs[blockIdx.y];
    const real sigma2 = (real)sigma2_d;
    const int nt64 = pd.ld / 64;
    int ti, tj;
    if (stage == -2) {
        // fused fit: only the diagonal 128 x 128 tiles (potrf and look-ahead read them from the slab); the tiles below are
        // evaluated by the factorisation at their first use.  Three 64 x 64 tiles per diagonal tile.
        const int c = blockIdx.x / 3, r = blockIdx.x - 3 * c;
        if (c >= pd.nt) return;
        ti = 2 * c + (r > 0);
        tj = 2 * c + (r > 1);
    } else if (stage < 0) {
        const int ntiles = nt64 * (nt64 + 1) / 2;
        const int t = blockIdx.x;
        if (t >= ntiles) return;
        ti = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
        while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
        while (ti * (ti + 1) / 2 > t) --ti;
        tj = t - ti * (ti + 1) / 2;
    } else {
        const int c = stage == 0 ? 0 : stage - (max_nt - pd.nt);       // block column of this patch at this stage
        if (c < (stage == 0 ? 0 : 1) || c >= pd.nt) return;
        tj = 2 * c + (blockIdx.x & 1);
        ti = tj + (int)(blockIdx.x >> 1);
        if (ti >= nt64) return;
    }
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i0 = ti * 64 + 4 * tx, j0 = tj * 64 + 4 * ty;
    const real *xs = x + pd.xoff;
    real xi[4][D], xj[4][D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const real4_t vi = *reinterpret_cast<const real4_t *>(xs + (int64_t)d * pd.ld + i0);
        const real4_t vj = *reinterpret_cast<const real4_t *>(xs + (int64_t)d * pd.ld + j0);
#pragma unroll
        for (int a = 0; a < 4; ++a) { xi[a][d] = vi[a]; xj[a][d] = vj[a]; }
    }
    real *S = A + pd.aoff;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int j = j0 + b;
        real4_t o;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int i = i0 + a;
            real v;
            if (i < pd.n && j < pd.n) {
                v = (i >= j) ? kern_eval<D, FAM, real>(th, xi[a], xj[b]) : kern_eval<D, FAM, real>(th, xj[b], xi[a]);
                if (i == j) {
                    if (dg) v = v + dg[pd.yoff + i];      // the kernel's own point-dependent diagonal term (pmk_model_set_diag)
                    v = v + sigma2;
                }
            } else {
                v = (i == j) ? (real)1 : (real)0;
            }
            o[a] = v;
        }
        *reinterpret_cast<real4_t *>(S + i0 + (int64_t)j * pd.ld) = o;
    }
}

template <int D>
static int launch_slab_D(const pmk_model *m, const pmk_kernel_desc &th, double sigma2, hipStream_t s, int64_t p0, int64_t np,
                         int stage)
{
    const int nt64 = m->max_nt * (TILE / 64);
    dim3 grid((unsigned)(stage == -2 ? 3 * m->max_nt : stage < 0 ? nt64 * (nt64 + 1) / 2 : 2 * nt64), (unsigned)np);
    if (th.family == PMK_SPLINE34)
        hipLaunchKernelGGL((kmat_slab_kernel<D, PMK_SPLINE34>), grid, dim3(256), 0, s, m->d_desc + p0, (const real *)m->d_x, (real *)m->d_a, th, sigma2, stage, m->max_nt, (const real *)m->d_diag);
    else
        hipLaunchKernelGGL((kmat_slab_kernel<D, 0>), grid, dim3(256), 0, s, m->d_desc + p0, (const real *)m->d_x, (real *)m->d_a, th, sigma2, stage, m->max_nt, (const real *)m->d_diag);
    PMK_HIP(hipGetLastError());
    return 0;
}

#define PMK_DISPATCH_D(D, CALL)                                    \
    switch (D) {                                                   \
    case 1: { constexpr int DD = 1; CALL; } break;                 \
    case 2: { constexpr int DD = 2; CALL; } break;                 \
    case 3: { constexpr int DD = 3; CALL; } break;                 \
    case 4: { constexpr int DD = 4; CALL; } break;                 \
    default: set_error("unsupported input dimension %d (1..%d)", (int)(D), MAX_D); return -2; \
    }

int launch_kernel_matrix_slabs(const pmk_model *m, const pmk_kernel_desc &th, double sigma2, hipStream_t s,
                               int64_t p0, int64_t np, int stage)
{
    int rc = 0;
    PMK_DISPATCH_D(m->D, rc = launch_slab_D<DD>(m, th, sigma2, s, p0, np, stage));
    return rc;
}

}  // namespace PMK_NS
}  // namespace pmk
