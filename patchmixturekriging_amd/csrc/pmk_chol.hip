// K2/K3: batched blocked Cholesky (left-looking, 128-wide block columns) + the two triangular
// solves for the GP weights.  Replaces, per patch, `cholesky(U)` and `c = U\y` of the reference
// (src/RKHS/mixtureGP.jl:106-112): one factorisation serves both.
//
// Step k of the factorisation (k = 0 .. nt-1), all patches at once, two launches:
//   chol_diag_kernel  : one workgroup per patch.  T = A[kk] - L[k,0:k] L[k,0:k]^T on MFMA, unblocked
//                       potrf of T in LDS -> L[kk]; inverse of L[kk] (used by every later TRSM as a
//                       GEMM); forward-substitution piece z_k = L[kk]^-1 (y_k - L[k,0:k] z_0:k).
//   chol_panel_kernel : grid over the block rows below.  T = A[i,k] - L[i,0:k] L[k,0:k]^T on MFMA
//                       (the contraction the north star prices), then L[i,k] = T L[kk]^-T as a second
//                       MFMA product with the accumulator tile reused in registers as the B operand.
// The slab is read once per block column (left-looking): reads only, no trailing-matrix
// read-modify-write.  chol_backsolve_kernel then gives c = L^-T z.
#include "pmk_mfma.h"

namespace pmk {

constexpr int LDT = TILE + 1;   // LDS leading dimension of the diagonal tile (row access conflict-free)
constexpr int PF_CHOL = 4;      // operand prefetch depth (k-steps) of the panel GEMM; must divide TILE/4
constexpr int PF_DIAG = 4;

// ---------------------------------------------------------------------------------------------
// diagonal block: GEMM update + potrf + inverse + forward-solve piece
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chol_diag_kernel(const PatchDesc *__restrict__ descs, double *__restrict__ A,
                                                        double *__restrict__ inv, const double *__restrict__ y,
                                                        double *__restrict__ z, int32_t *__restrict__ info, int k)
{
    const PatchDesc pd = descs[blockIdx.x];
    if (k >= pd.nt) return;
    __shared__ double T[TILE * LDT];
    __shared__ double dinv[TILE];
    __shared__ double rhs[2 * TILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = wave >> 1, g = wave & 1;          // 64-row half, 64-column half of the tile
    double *S = A + pd.aoff;
    const int64_t ld = pd.ld;
    const int K = k * TILE;
    const int64_t r0 = (int64_t)k * TILE + 64 * h, c0 = (int64_t)k * TILE + 64 * g;

    // ---- T = A[kk] - L[k,0:k] L[k,0:k]^T  (lower 64x64 sub-tiles only)
    if (!(h == 0 && g == 1)) {
        WaveTile<2, 2> acc;
        acc.zero();
        if (K > 0) gemm_nt<2, 2, PF_DIAG>(acc, S + c0, ld, S + r0, ld, K, lane);
#pragma unroll
        for (int fi = 0; fi < 4; ++fi)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int pj = 0; pj < 2; ++pj) {
                    const int cl = 64 * g + 32 * (fi >> 1) + 2 * ((lane >> 4) + 4 * q) + (fi & 1);
                    const int rl = 64 * h + 32 * pj + 2 * (lane & 15);
                    const double2_t a =
                        *reinterpret_cast<const double2_t *>(S + (int64_t)k * TILE + rl + ((int64_t)k * TILE + cl) * ld);
                    T[rl + cl * LDT] = a[0] - acc.f[fi][2 * pj][q];
                    T[rl + 1 + cl * LDT] = a[1] - acc.f[fi][2 * pj + 1][q];
                }
    }
    // ---- forward-substitution partial sums: rhs = L[k,0:k] z_0:k  (two column halves)
    {
        const int row = tid & 127, half = tid >> 7;
        double s = 0.0;
        const double *Lr = S + (int64_t)k * TILE + row;
        const double *zz = z + pd.yoff;
        const int cbeg = half * (K / 2), cend = cbeg + K / 2;
        for (int c = cbeg; c < cend; ++c) s += Lr[(int64_t)c * ld] * zz[c];
        rhs[tid] = s;
    }
    __syncthreads();

    // ---- unblocked right-looking potrf on the lower triangle of T
    int bad = 0;
    for (int j = 0; j < TILE; ++j) {
        double d = T[j + j * LDT];
        if (!(d > 0.0)) {          // not positive definite (or NaN): record the leading minor, keep going
            if (!bad) bad = k * TILE + j + 1;
            d = 1.0;
        }
        const double s = sqrt(d);
        __syncthreads();
        if (tid == j) T[j + j * LDT] = s;
        if (tid > j && tid < TILE) T[tid + j * LDT] = T[tid + j * LDT] / s;
        __syncthreads();
        const int i = tid & 127;
        if (i > j) {
            const double lij = T[i + j * LDT];
            for (int c = j + 1 + (tid >> 7); c <= i; c += 2) T[i + c * LDT] -= lij * T[c + j * LDT];
        }
        __syncthreads();
    }
    if (bad && tid == 0 && info[blockIdx.x] == 0) info[blockIdx.x] = bad;

    // ---- L[kk] -> slab (lower; the strict upper part of the slab block is zeroed)
    for (int e = tid; e < TILE * TILE; e += 256) {
        const int i = e & 127, c = e >> 7;
        S[(int64_t)k * TILE + i + ((int64_t)k * TILE + c) * ld] = (i >= c) ? T[i + c * LDT] : 0.0;
    }
    // ---- inverse of L[kk], one column per thread; X^T is kept in the strict upper triangle of T
    if (tid < TILE) {
        const int c = tid;
        const double xcc = 1.0 / T[c + c * LDT];
        dinv[c] = xcc;
        for (int i = c + 1; i < TILE; ++i) {
            double s = T[i + c * LDT] * xcc;
            for (int kk = c + 1; kk < i; ++kk) s += T[i + kk * LDT] * T[c + kk * LDT];
            T[c + i * LDT] = -s / T[i + i * LDT];
        }
    }
    __syncthreads();
    double *Li = inv + pd.ioff + (int64_t)k * TILE * TILE;
    for (int e = tid; e < TILE * TILE; e += 256) {
        const int i = e & 127, c = e >> 7;
        Li[e] = (i > c) ? T[c + i * LDT] : (i == c ? dinv[c] : 0.0);
    }
    // ---- z_k = L[kk]^-1 (y_k - rhs)
    if (tid < TILE) rhs[tid] = y[pd.yoff + (int64_t)k * TILE + tid] - (rhs[tid] + rhs[tid + TILE]);
    __syncthreads();
    if (tid < TILE) {
        const int i = tid;
        double s = dinv[i] * rhs[i];
        for (int c = 0; c < i; ++c) s += T[c + i * LDT] * rhs[c];
        z[pd.yoff + (int64_t)k * TILE + i] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// block column below the diagonal: MFMA update + in-register TRSM
// workgroup = one 128-row tile = 4 waves x (32 rows x 128 columns); the waves are independent (no
// LDS, no barrier) and two workgroups share a CU (2 waves per SIMD hide each other's load latency)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void chol_panel_kernel(const PatchDesc *__restrict__ descs, double *__restrict__ A,
                                                            const double *__restrict__ inv, int k)
{
    const PatchDesc pd = descs[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)(k + 1 + blockIdx.x) * TILE + 32 * wave;
    if (r0 >= pd.ld) return;
    double *S = A + pd.aoff;
    const int64_t ld = pd.ld;
    const int64_t c0 = (int64_t)k * TILE;
    double *out = S + r0 + 2 * (lane & 15) + (c0 + 2 * (lane >> 4)) * ld;   // element (fi = 0, q = 0)

    // acc starts as -A[rows, block column k]; the GEMM adds L[rows,0:k] L[k,0:k]^T, so acc = -T
    WaveTile<4, 1> acc;   // I = the 128 columns of block column k, J = 32 rows
#pragma unroll
    for (int fi = 0; fi < 8; ++fi)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cl = 32 * (fi >> 1) + 8 * q + (fi & 1);
            const double2_t a = *reinterpret_cast<const double2_t *>(out + cl * ld);
            acc.f[fi][0][q] = -a[0];
            acc.f[fi][1][q] = -a[1];
        }
    if (k > 0) gemm_nt<4, 1, PF_CHOL>(acc, S + c0, ld, S + r0, ld, k * TILE, lane);
    // -L[rows, k] = (-T) L[kk]^-T  :  out[c'][r] = sum_c Linv[c'][c] (-T)[c][r]
    tri_solve_inplace<1>(acc, inv + pd.ioff + (int64_t)k * TILE * TILE, lane);
#pragma unroll
    for (int fi = 0; fi < 8; ++fi)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cl = 32 * (fi >> 1) + 8 * q + (fi & 1);
            double2_t o;
            o[0] = -acc.f[fi][0][q];
            o[1] = -acc.f[fi][1][q];
            *reinterpret_cast<double2_t *>(out + cl * ld) = o;
        }
}

// ---------------------------------------------------------------------------------------------
// c = L^-T z, one workgroup per patch, block rows from the last to the first
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chol_backsolve_kernel(const PatchDesc *__restrict__ descs,
                                                             const double *__restrict__ A, const double *__restrict__ inv,
                                                             const double *__restrict__ z, double *__restrict__ cvec)
{
    const PatchDesc pd = descs[blockIdx.x];
    __shared__ double part[TILE];
    __shared__ double r[TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double *S = A + pd.aoff;
    const int64_t ld = pd.ld;
    double *c = cvec + pd.yoff;
    for (int k = pd.nt - 1; k >= 0; --k) {
        // part[col] = sum_{i >= (k+1)*TILE} L[i, col] c[i]   for the 128 columns of block k
        const int64_t i0 = (int64_t)(k + 1) * TILE;
        for (int cc = wave; cc < TILE; cc += 4) {
            const double *col = S + ((int64_t)k * TILE + cc) * ld;
            double s = 0.0;
            for (int64_t i = i0 + lane; i < ld; i += 64) s += col[i] * c[i];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0) part[cc] = s;
        }
        __syncthreads();
        if (tid < TILE) r[tid] = z[pd.yoff + (int64_t)k * TILE + tid] - part[tid];
        __syncthreads();
        if (tid < TILE) {
            // c_k[col] = sum_{i >= col} Linv[i][col] r[i]
            const double *Li = inv + pd.ioff + (int64_t)k * TILE * TILE + (int64_t)tid * TILE;
            double s = 0.0;
            for (int i = tid; i < TILE; ++i) s += Li[i] * r[i];
            c[(int64_t)k * TILE + tid] = s;
        }
        __threadfence_block();
        __syncthreads();
    }
}

int launch_cholesky(pmk_model *m, hipStream_t s)
{
    // gemm_nt consumes K in groups of 4*PF k-indices; K is always a multiple of TILE here
    static_assert(TILE % (4 * PF_DIAG) == 0 && TILE % (4 * PF_CHOL) == 0, "prefetch depth must divide TILE/4");
    PMK_HIP(hipMemsetAsync(m->d_info, 0, sizeof(int32_t) * m->P, s));
    m->ctx->panel_n = 0;
    for (int k = 0; k < m->max_nt; ++k) {
        hipLaunchKernelGGL(chol_diag_kernel, dim3((unsigned)m->P), dim3(256), 0, s, m->d_desc, m->d_a, m->d_inv,
                           m->d_y, m->d_z, m->d_info, k);
        const int below = m->max_nt - k - 1;
        if (below > 0) {
            pmk_ctx *c = m->ctx;
            const bool fine = c->timers >= 2;
            if (fine) {
                while ((int)c->panel_ev.size() <= k) {
                    hipEvent_t a, b;
                    PMK_HIP(hipEventCreate(&a));
                    PMK_HIP(hipEventCreate(&b));
                    c->panel_ev.push_back({a, b});
                }
                PMK_HIP(hipEventRecord(c->panel_ev[(size_t)k].first, s));
            }
            hipLaunchKernelGGL(chol_panel_kernel, dim3((unsigned)below, (unsigned)m->P), dim3(256), 0, s,
                               m->d_desc, m->d_a, m->d_inv, k);
            if (fine) {
                PMK_HIP(hipEventRecord(c->panel_ev[(size_t)k].second, s));
                c->panel_n = k + 1;
            }
        }
    }
    PMK_HIP(hipGetLastError());
    return 0;
}

int launch_backsolve(pmk_model *m, hipStream_t s)
{
    hipLaunchKernelGGL(chol_backsolve_kernel, dim3((unsigned)m->P), dim3(256), 0, s, m->d_desc, m->d_a, m->d_inv,
                       m->d_z, m->d_c);
    PMK_HIP(hipGetLastError());
    return 0;
}

}  // namespace pmk
