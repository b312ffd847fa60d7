// K2/K3: batched blocked Cholesky (left-looking, 128-wide block columns) + the two triangular
// solves for the GP weights.  Replaces, per patch, `cholesky(U)` and `c = U\y` of the reference
// (src/RKHS/mixtureGP.jl:106-112): one factorisation serves both.
//
// Step k of the factorisation (k = 0 .. nt-1), all patches at once, two launches:
//   chol_diag_kernel  : one workgroup per patch.  Applies the last block column to the diagonal tile
//                       (A[kk] -= L[k,k-1] L[k,k-1]^T, MFMA, in place in the slab), factors it with the tile
//                       distributed over the registers of the 256 threads (right-looking, two barriers per
//                       pivot), stores L[kk] and the negated inverses of its four 32 x 32 diagonal blocks
//                       (every later TRSM is MFMA block substitution with them); the right-hand side rides
//                       along as an extra row, so z_k = L[kk]^-1 (y_k - L[k,0:k] z_0:k) falls out of the
//                       same elimination.
//   chol_panel_kernel : grid over the block rows below + one "look-ahead" workgroup per patch.
//                       Block rows: T = A[i,k] - L[i,0:k] L[k,0:k]^T on MFMA (the contraction the north
//                       star prices), then L[i,k] = T L[kk]^-T by in-register block substitution.
//                       Look-ahead workgroup: applies block columns 0..k-1 to the NEXT diagonal tile
//                       (A[k+1,k+1] -= L[k+1,0:k] L[k+1,0:k]^T) and to the next forward-solve piece, so
//                       the serial diagonal kernel only ever sees one block column of GEMM.
// The slab is read once per block column (left-looking): reads only, no trailing-matrix
// read-modify-write.  chol_backsolve_kernel then gives c = L^-T z.
#include <cstdlib>

#include "pmk_mfma.h"

namespace pmk {
namespace PMK_NS {

constexpr int PF_CHOL = 4;      // I-operand prefetch depth (k-steps) of the panel GEMM; must divide TILE/4
#ifndef PMK_PFJ
#define PMK_PFJ 4
#endif
constexpr int PFJ_CHOL = PMK_PFJ;   // J-operand (own rows, HBM) prefetch depth
constexpr int PF_DIAG = 4;
constexpr int SB = 32;          // sub-block of the in-LDS potrf and of the TRSM block substitution

// Fused kernel-matrix build (K1 folded into the factorisation): entry (i, j) of U = K + sigma2 I of a
// patch, evaluated from the resident coordinates exactly as constructkernelmatrix! + the diagonal update
// do (row point first for i >= j, mirrored above; src/RKHS/RKHS.jl:21-31, mixtureGP.jl:102-104), with the
// identity padding of the slab.  FUSE = 0 reads the value kmat_slab_kernel wrote instead.
template <int D, int FAM>
struct TileSource {
    const real *xs;     // SoA coordinates of the patch
    int64_t ld;
    int n;
    real sigma2;
    pmk_kernel_desc th;
    __device__ __forceinline__ void point(int i, real *p) const
    {
#pragma unroll
        for (int d = 0; d < D; ++d) p[d] = xs[(int64_t)d * ld + i];
    }
    __device__ __forceinline__ real value(int i, const real *pi, int j, const real *pj) const
    {
        real v = (i >= j) ? kern_eval<D, FAM, real>(th, pi, pj) : kern_eval<D, FAM, real>(th, pj, pi);
        v = (i == j) ? v + sigma2 : v;
        const bool inside = i < n && j < n;
        return inside ? v : ((i == j) ? 1.0 : 0.0);
    }
};

// ---------------------------------------------------------------------------------------------
// diagonal block
// ---------------------------------------------------------------------------------------------
template <int D, int FAM, int FUSE>
__global__ __launch_bounds__(256, 2) void chol_diag_kernel(const PatchDesc *__restrict__ descs, real *__restrict__ A,
                                                           real *__restrict__ ninv, const real *__restrict__ y,
                                                           const real *__restrict__ ytmp, real *__restrict__ z,
                                                           int32_t *__restrict__ info, int k,
                                                           const real *__restrict__ x, pmk_kernel_desc th, double sigma2,
                                                           int skip)
{
    const PatchDesc pd = descs[blockIdx.x];
    if (k >= pd.nt) return;
    // LDS stays small (~38 KB, 256 registers): the tile itself lives in the slab (L2) and in registers, so a
    // diagonal workgroup can share a CU with a panel workgroup of another sub-batch (PMK_FIT_GROUPS)
    __shared__ real dblk[4][SB][SB + 1];     // the four 32 x 32 diagonal blocks of L[kk], for their inversion
    __shared__ real col[TILE + 1];           // scaled pivot column; col[TILE] = the forward-solve entry z_j
    __shared__ real rhs[2 * TILE];
    __shared__ real sdiag;
    __shared__ int s_bad;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = wave >> 1, g = wave & 1;          // 64-row half, 64-column half of the tile
    real *S = A + pd.aoff;
    const int64_t ld = pd.ld;
    const int64_t d0 = (int64_t)k * TILE;
    real *Akk = S + d0 + d0 * ld;
    if (tid == 0) s_bad = 0;

    // ---- A[kk] -= L[k,k-1] L[k,k-1]^T in place in the slab (older block columns were applied by the look-ahead
    //      workgroup of the previous panel launch); lower 64 x 64 sub-tiles only
    if (!(h == 0 && g == 1) && ((k > 0 && !(skip & 16)) || (FUSE && k == 0))) {
        // acc starts as -A[kk] (all of the sub-tile's loads in flight at once; read-modify-write per element made
        // hipcc serialise 16 L2 round trips behind the GEMM), the GEMM adds L L^T, the store negates
        WaveTile<2, 2> acc;
#pragma unroll
        for (int fi = 0; fi < 4; ++fi)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int pj = 0; pj < 2; ++pj) {
                    const int cl = 64 * g + tile_i(fi, lane, q);
                    const int rl = 64 * h + 32 * pj + 2 * (lane & 15);
                    real2_t a;
                    if (FUSE && k == 0) {          // first tile: nothing was written to the slab, evaluate K here
                        const TileSource<D, FAM> src{x + pd.xoff, ld, pd.n, (real)sigma2, th};
                        real pc[D], pr0[D], pr1[D];
                        src.point(cl, pc); src.point(rl, pr0); src.point(rl + 1, pr1);
                        a[0] = src.value(rl, pr0, cl, pc);
                        a[1] = src.value(rl + 1, pr1, cl, pc);
                    } else {
                        a = *reinterpret_cast<const real2_t *>(Akk + rl + (int64_t)cl * ld);
                    }
                    acc.f[fi][2 * pj][q] = -a[0];
                    acc.f[fi][2 * pj + 1][q] = -a[1];
                }
        if (k > 0) {
            const real *Lk = S + d0 + (d0 - TILE) * ld;      // L[k, k-1]: 128 x 128
            gemm_nt<2, 2, PF_DIAG>(acc, Lk + 64 * g, ld, Lk + 64 * h, ld, TILE, lane);
        }
#pragma unroll
        for (int fi = 0; fi < 4; ++fi)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int pj = 0; pj < 2; ++pj) {
                    const int cl = 64 * g + tile_i(fi, lane, q);
                    const int rl = 64 * h + 32 * pj + 2 * (lane & 15);
                    real2_t a;
                    a[0] = -acc.f[fi][2 * pj][q];
                    a[1] = -acc.f[fi][2 * pj + 1][q];
                    *reinterpret_cast<real2_t *>(Akk + rl + (int64_t)cl * ld) = a;
                }
    }
    // ---- forward-solve right-hand side: y_k - L[k,0:k-1] z (look-ahead) - L[k,k-1] z_{k-1}
    {
        const int row = tid & 127, half = tid >> 7;
        real s = 0.0;
        if (k > 0) {
            const real *Lr = S + d0 + row + (d0 - TILE + 64 * half) * ld;
            const real *zz = z + pd.yoff + d0 - TILE + 64 * half;
            for (int cb = 0; cb < 64; cb += 16) {          // 16 independent column loads in flight per batch
                real lv[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) lv[c] = Lr[(int64_t)(cb + c) * ld];
#pragma unroll
                for (int c = 0; c < 16; ++c) s += lv[c] * zz[cb + c];
            }
        }
        rhs[tid] = s;
    }
    __threadfence_block();
    __syncthreads();
    if (tid < TILE) {
        const real base = (k == 0) ? y[pd.yoff + tid] : ytmp[pd.yoff + d0 + tid];
        rhs[tid] = base - (rhs[tid] + rhs[tid + TILE]);
    }
    __syncthreads();

    // ---- potrf of the tile: right-looking, the tile distributed over the registers of all 256 threads
    //      (thread (tr, tc) of a 16 x 16 grid holds A[tr + 16 a][tc + 16 b], a, b < 8; the threads with tr == 0
    //      also carry the right-hand side as an extra row, which the elimination turns into z_k = L[kk]^-1 rhs).
    //      Per pivot: the 16 threads that own column j scale it and publish it through LDS, one barrier, every
    //      thread applies the rank-1 update to its own entries, one barrier.  The pivot's reciprocal square root
    //      comes from v_rsq + Newton steps: the 128 pivots are a serial latency chain, this is its critical path.
    {
        const int tr = tid & 15, tc = tid >> 4;
        real a_[8][8], rr[8];
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b2 = 0; b2 < 8; ++b2)
                a_[a][b2] = (b2 <= a) ? Akk[(tr + 16 * a) + (int64_t)(tc + 16 * b2) * ld] : (real)0;
#pragma unroll
        for (int b2 = 0; b2 < 8; ++b2) rr[b2] = rhs[tc + 16 * b2];
        if (tid == 0) sdiag = a_[0][0];
        __syncthreads();
        // two-level pivot loop: the 16-column block index bj is a compile-time constant in each copy of the body, so
        // a_[.][bj] is a static register reference and the update loops run over exactly the live blocks -- no
        // per-pivot scalar branches (an earlier version guarded every 16 x 16 block with a runtime test: ~40
        // s_cbranch per pivot cost more than the arithmetic they skipped)
#pragma unroll
        for (int bj = 0; bj < 8; ++bj) {
            for (int jj = 0; jj < ((skip & 1) ? 0 : 16); ++jj) {
                const int j = 16 * bj + jj;
                if (tc == jj) {
                    real d = sdiag;
                    if (!(d > (real)0)) {                     // not positive definite (or NaN): record, keep going
                        if (tr == 0 && s_bad == 0) s_bad = k * TILE + j + 1;
                        d = 1;
                    }
                    const real rs = rsqrt_real(d);
                    const real ljj = d * rs;
#pragma unroll
                    for (int a = bj; a < 8; ++a) {
                        const int r = tr + 16 * a;
                        const real v = (r > j) ? a_[a][bj] * rs : ((r == j) ? ljj : a_[a][bj]);
                        a_[a][bj] = v;
                        col[r] = (r > j) ? v : (real)0;
                    }
                    if (tr == 0) {                            // the extra row: z_j = rhs_j / L[j][j]
                        rr[bj] = rr[bj] * rs;
                        col[TILE] = rr[bj];
                    }
                }
                __syncthreads();
                real cr[8], cc[8];
#pragma unroll
                for (int a = bj; a < 8; ++a) { cr[a] = col[tr + 16 * a]; cc[a] = col[tc + 16 * a]; }
#pragma unroll
                for (int a = bj; a < 8; ++a)
#pragma unroll
                    for (int b2 = bj; b2 <= a; ++b2) a_[a][b2] -= cr[a] * cc[b2];   // col[] is 0 for rows <= j
                if (tr == 0) {
                    const real zj = col[TILE];
#pragma unroll
                    for (int b2 = bj; b2 < 8; ++b2) rr[b2] -= zj * cc[b2];
                }
                // publish the next pivot's diagonal entry: A[j+1][j+1] lives in block (bj, bj), or in block
                // (bj+1, bj+1) of thread (0, 0) when j+1 starts the next block
                if (jj < 15) {
                    if (tr == jj + 1 && tc == jj + 1) sdiag = a_[bj][bj];
                } else if (bj < 7) {
                    if (tid == 0) sdiag = a_[bj + 1 < 8 ? bj + 1 : 7][bj + 1 < 8 ? bj + 1 : 7];
                }
                __syncthreads();
            }
        }
        // ---- L[kk] -> slab (lower; the strict upper part of the slab block is zeroed), the four diagonal
        //      32 x 32 blocks -> LDS for their inversion, z_k -> global
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b2 = 0; b2 < 8; ++b2) {
                const int r = tr + 16 * a, c = tc + 16 * b2;
                const real v = (b2 <= a && r >= c) ? a_[a][b2] : (real)0;
                if (!(skip & 4)) Akk[r + (int64_t)c * ld] = v;
                if ((r >> 5) == (c >> 5)) dblk[r >> 5][r & 31][c & 31] = v;
            }
        if (tr == 0) {
#pragma unroll
            for (int b2 = 0; b2 < 8; ++b2) z[pd.yoff + d0 + tc + 16 * b2] = rr[b2];
        }
    }
    __syncthreads();
    if (tid == 0 && s_bad && info[blockIdx.x] == 0) info[blockIdx.x] = s_bad;
    // ---- negated inverses of the four 32 x 32 diagonal blocks: wave w inverts block w, one thread per column
    if (lane < SB && !(skip & 2)) {
        const int c = lane;
        real xcol[SB];
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            real sacc = (i == c) ? (real)1 : (real)0;
#pragma unroll
            for (int l = 0; l < i; ++l) sacc -= dblk[wave][i][l] * xcol[l];
            xcol[i] = sacc / dblk[wave][i][i];
        }
        real *Ni = ninv + pd.ioff + (int64_t)k * (4 * SB * SB) + (int64_t)wave * (SB * SB);
#pragma unroll
        for (int i = 0; i < SB; ++i) Ni[i + SB * c] = (i >= c) ? -xcol[i] : (real)0;
    }
}

// ---------------------------------------------------------------------------------------------
// block column below the diagonal: MFMA update + in-register TRSM, plus the look-ahead workgroup
// block-row workgroup = one 128-row tile = 4 waves x (32 rows x 128 columns); the waves are
// independent (no LDS, no barrier) and two workgroups share a CU (2 waves per SIMD)
// ---------------------------------------------------------------------------------------------
template <int D, int FAM, int FUSE>
__global__ __launch_bounds__(256, 2) void chol_panel_kernel(const PatchDesc *__restrict__ descs, real *__restrict__ A,
                                                            const real *__restrict__ ninv, const real *__restrict__ y,
                                                            const real *__restrict__ z, real *__restrict__ ytmp, int k,
                                                            const real *__restrict__ x, pmk_kernel_desc th, double sigma2)
{
    // logical (block row, patch) from the XCD-aware id: all workgroups of a patch land on one XCD and
    // share the I-operand (block row k of L) through that XCD's L2
    const int lid = xcd_remap(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x * gridDim.y);
    const int bx = lid % gridDim.x;
    const PatchDesc pd = descs[lid / gridDim.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    real *S = A + pd.aoff;
    const int64_t ld = pd.ld;
    const int64_t c0 = (int64_t)k * TILE;

    if (bx == (int)gridDim.x - 1) {
        // ---- look-ahead for diagonal tile k+1: block columns 0..k-1 (column k is being produced by
        //      this very launch and is applied by the next diagonal kernel)
        const int64_t t0 = c0 + TILE;
        if (k + 1 >= pd.nt) return;
        const int h = wave >> 1, g = wave & 1;
        if ((k > 0 || FUSE) && !(h == 0 && g == 1)) {
            // acc = -A[k+1,k+1] (or -K evaluated) up-front, GEMM adds L L^T, the store negates (see diag_kernel)
            WaveTile<2, 2> acc;
            real *Att = S + t0 + t0 * ld;
            const TileSource<D, FAM> src{x + pd.xoff, ld, pd.n, (real)sigma2, th};
#pragma unroll
            for (int fi = 0; fi < 4; ++fi)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int pj = 0; pj < 2; ++pj) {
                        const int cl = 64 * g + tile_i(fi, lane, q);
                        const int rl = 64 * h + 32 * pj + 2 * (lane & 15);
                        real2_t a;
                        if (FUSE) {
                            real pc[D], pr0[D], pr1[D];
                            const int gc = (int)t0 + cl, gr = (int)t0 + rl;
                            src.point(gc, pc); src.point(gr, pr0); src.point(gr + 1, pr1);
                            a[0] = src.value(gr, pr0, gc, pc);
                            a[1] = src.value(gr + 1, pr1, gc, pc);
                        } else {
                            a = *reinterpret_cast<const real2_t *>(Att + rl + (int64_t)cl * ld);
                        }
                        acc.f[fi][2 * pj][q] = -a[0];
                        acc.f[fi][2 * pj + 1][q] = -a[1];
                    }
            if (k > 0) gemm_nt<2, 2, PF_DIAG>(acc, S + t0 + 64 * g, ld, S + t0 + 64 * h, ld, k * TILE, lane);
#pragma unroll
            for (int fi = 0; fi < 4; ++fi)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int pj = 0; pj < 2; ++pj) {
                        const int cl = 64 * g + tile_i(fi, lane, q);
                        const int rl = 64 * h + 32 * pj + 2 * (lane & 15);
                        real2_t a;
                        a[0] = -acc.f[fi][2 * pj][q];
                        a[1] = -acc.f[fi][2 * pj + 1][q];
                        *reinterpret_cast<real2_t *>(Att + rl + (int64_t)cl * ld) = a;
                    }
        }
        if (h == 0 && g == 1) {
            // the wave without a GEMM sub-tile does the forward-solve piece: two rows per lane
            const real *Lr = S + t0 + 2 * lane;
            const real *zz = z + pd.yoff;
            real2_t s0 = {0.0, 0.0}, s1 = {0.0, 0.0}, s2 = {0.0, 0.0}, s3 = {0.0, 0.0};
            for (int c = 0; c < k * TILE; c += 16) {     // 16 independent column loads in flight per batch
                real2_t av[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) av[u] = *reinterpret_cast<const real2_t *>(Lr + (int64_t)(c + u) * ld);
#pragma unroll
                for (int u = 0; u < 16; u += 4) {
                    s0 += av[u] * zz[c + u]; s1 += av[u + 1] * zz[c + u + 1];
                    s2 += av[u + 2] * zz[c + u + 2]; s3 += av[u + 3] * zz[c + u + 3];
                }
            }
            const real2_t sum = (s0 + s1) + (s2 + s3);
            const real2_t yy = *reinterpret_cast<const real2_t *>(y + pd.yoff + t0 + 2 * lane);
            *reinterpret_cast<real2_t *>(ytmp + pd.yoff + t0 + 2 * lane) = yy - sum;
        }
        return;
    }

    // the TRSM operands (off-diagonal 32-blocks of L[kk] and the four -D^-1 blocks) go to LDS once per workgroup
    if ((int64_t)(k + 1 + bx) * TILE >= pd.ld) return;      // whole workgroup: ld is a multiple of TILE
    __shared__ real tri[TRI_LDS_DOUBLES];
    const int64_t r0 = (int64_t)(k + 1 + bx) * TILE + 32 * wave;
    real *out = S + r0 + 2 * (lane & 15) + c0 * ld;   // rows of this lane, first column of the block column

    // acc starts as -A[rows, block column k]; the GEMM adds L[rows,0:k] L[k,0:k]^T, so acc = -T.  The tile comes
    // from HBM (first touch): its loads are issued before the operand staging so that the two latencies overlap
    // instead of adding up.
    WaveTile<4, 1> acc;   // I = the 128 columns of block column k, J = 32 rows
    if (FUSE) {
        stage_tri_operands(tri, S + c0 + c0 * ld, ld, ninv + pd.ioff + (int64_t)k * (4 * SB * SB), tid, 256);
        const TileSource<D, FAM> src{x + pd.xoff, ld, pd.n, (real)sigma2, th};
        const int gr = (int)r0 + 2 * (lane & 15);
        real pr0[D], pr1[D];
        src.point(gr, pr0); src.point(gr + 1, pr1);
#pragma unroll
        for (int fi = 0; fi < 8; ++fi)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gc = (int)c0 + tile_i(fi, lane, q);
                real pc[D];
                src.point(gc, pc);
                acc.f[fi][0][q] = -src.value(gr, pr0, gc, pc);
                acc.f[fi][1][q] = -src.value(gr + 1, pr1, gc, pc);
            }
    } else {
#pragma unroll
        for (int fi = 0; fi < 8; ++fi)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cl = tile_i(fi, lane, q);
                const real2_t a = *reinterpret_cast<const real2_t *>(out + cl * ld);
                acc.f[fi][0][q] = a[0];
                acc.f[fi][1][q] = a[1];
            }
        __builtin_amdgcn_sched_barrier(0);      // keep the tile loads ahead of the staging loads
        stage_tri_operands(tri, S + c0 + c0 * ld, ld, ninv + pd.ioff + (int64_t)k * (4 * SB * SB), tid, 256);
#pragma unroll
        for (int fi = 0; fi < 8; ++fi)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc.f[fi][0][q] = -acc.f[fi][0][q];
                acc.f[fi][1][q] = -acc.f[fi][1][q];
            }
    }
    __syncthreads();
    if (k > 0) gemm_nt<4, 1, PF_CHOL, PFJ_CHOL>(acc, S + c0, ld, S + r0, ld, k * TILE, lane);
    // L[rows, k]^T = L[kk]^-1 T^T = -L[kk]^-1 (-T)^T : exactly what the block substitution returns
    tri_solve_inplace<1>(acc, tri, lane);
#pragma unroll
    for (int fi = 0; fi < 8; ++fi)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cl = tile_i(fi, lane, q);
            real2_t o;
            o[0] = acc.f[fi][0][q];
            o[1] = acc.f[fi][1][q];
            *reinterpret_cast<real2_t *>(out + cl * ld) = o;
        }
}

// ---------------------------------------------------------------------------------------------
// c = L^-T z, one workgroup (16 waves) per patch, block rows from the last to the first
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void chol_backsolve_kernel(const PatchDesc *__restrict__ descs,
                                                              const real *__restrict__ A, const real *__restrict__ ninv,
                                                              const real *__restrict__ z, real *__restrict__ cvec,
                                                              int cs_in_lds)
{
    const PatchDesc pd = descs[blockIdx.x];
    extern __shared__ double sm_raw[];
    real *sm = reinterpret_cast<real *>(sm_raw);
    // LDS: the solution so far (ld), the right-hand side of the current block (TILE), the current diagonal tile
    // L[kk] (padded leading dimension: the substitution walks columns, one thread per column) and its four
    // negated inverted 32 x 32 blocks.  The tile used to be read from the slab inside the substitution: 32 threads
    // chasing strided global loads cost ~60 us per block row, most of this kernel's time.
    constexpr int LDN = SB + 1;
    real *r = sm;
    real *Lt = r + TILE;                 // the six 32 x 32 blocks of L[kk] below its block diagonal, ld 33
    real *Nt = Lt + 6 * SB * LDN;
    real *cs = cs_in_lds ? Nt + 4 * SB * LDN : cvec + pd.yoff;     // very long patches keep c in global memory
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const real *S = A + pd.aoff;
    const int64_t ld = pd.ld;
    for (int k = pd.nt - 1; k >= 0; --k) {
        const int64_t d0 = (int64_t)k * TILE;
        // stage L[kk] and its -D^-1 blocks (coalesced), overlapping with the column dots below
        {
            // all ten loads of a thread in flight before the first LDS store (a rolled load -> store loop is one
            // L2 round trip per piece)
            const real *Lkk = S + d0 + d0 * ld;
            const real *Ni = ninv + pd.ioff + (int64_t)k * (4 * SB * SB);
            real lv[6], nv[4];
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int e = tid + 1024 * u;
                const int b = e >> 10, i = e & 31, c = (e >> 5) & 31;
                const int bi = (b >= 3) ? 3 : (b >= 1 ? 2 : 1), bs = b - bi * (bi - 1) / 2;   // block (bi, bs), bi > bs
                lv[u] = Lkk[SB * bi + i + (int64_t)(SB * bs + c) * ld];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) nv[u] = Ni[tid + 1024 * u];
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int e = tid + 1024 * u;
                const int b = e >> 10, i = e & 31, c = (e >> 5) & 31;
                Lt[b * (SB * LDN) + i + c * LDN] = lv[u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = tid + 1024 * u;
                const int sb = e >> 10, i = e & 31, c = (e >> 5) & 31;
                Nt[sb * (SB * LDN) + i + c * LDN] = nv[u];
            }
        }
        // r[col] = z_k[col] - sum_{i >= d0 + TILE} L[i, d0 + col] c[i]: one wave per column, coalesced rows
        const int64_t i0 = d0 + TILE;
        for (int cc = wave; cc < TILE; cc += 16) {
            const real *col = S + (d0 + cc) * ld;
            real s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int64_t i = i0 + lane;
            for (; i + 192 < ld; i += 256) {
                s0 += col[i] * cs[i];
                s1 += col[i + 64] * cs[i + 64];
                s2 += col[i + 128] * cs[i + 128];
                s3 += col[i + 192] * cs[i + 192];
            }
            for (; i < ld; i += 64) s0 += col[i] * cs[i];
            real s = (s0 + s1) + (s2 + s3);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0) r[cc] = z[pd.yoff + d0 + cc] - s;
        }
        __syncthreads();
        // c_k = L[kk]^-T r by block backward substitution with the negated inverted 32 x 32 blocks, all from LDS
        for (int s = 3; s >= 0; --s) {
            if (tid < SB) {
                const int col = SB * s + tid;
                real v = r[col];
                for (int bi = s + 1; bi < 4; ++bi) {
                    const real *blk = Lt + (bi * (bi - 1) / 2 + s) * (SB * LDN) + tid * LDN;
                    for (int i = 0; i < SB; ++i) v -= blk[i] * cs[d0 + SB * bi + i];
                }
                r[col] = v;
            }
            __syncthreads();
            if (tid < SB) {
                real v = 0.0;
                for (int i = tid; i < SB; ++i) v -= Nt[s * (SB * LDN) + i + tid * LDN] * r[SB * s + i];   // D^-T r
                cs[d0 + SB * s + tid] = v;
            }
            __threadfence_block();
            __syncthreads();
        }
    }
    if (cs_in_lds)
        for (int i = tid; i < pd.ld; i += 1024) cvec[pd.yoff + i] = cs[i];
}

template <int D, int FAM, int FUSE>
static int launch_cholesky_T(pmk_model *m, hipStream_t s, int64_t p0, int64_t np)
{
    // gemm_nt consumes K in groups of 4*PF k-indices; K is always a multiple of TILE here
    static_assert(TILE % (4 * PF_DIAG) == 0 && TILE % (4 * PF_CHOL) == 0, "prefetch depth must divide TILE/4");
    PMK_HIP(hipMemsetAsync(m->d_info + p0, 0, sizeof(int32_t) * np, s));
    pmk_ctx *c = m->ctx;
    c->panel_n = 0;
    real *ytmp = (real *)m->d_c;      // the weight vector is free until the back substitution: scratch for y - L z
    static const int dbg_skip = getenv("PMK_DBG_DIAG_SKIP") ? atoi(getenv("PMK_DBG_DIAG_SKIP")) : 0;   // timing experiments only
    for (int k = 0; k < m->max_nt; ++k) {
        hipLaunchKernelGGL((chol_diag_kernel<D, FAM, FUSE>), dim3((unsigned)np), dim3(256), 0, s, m->d_desc + p0, (real *)m->d_a,
                           (real *)m->d_inv, (real *)m->d_y, ytmp, (real *)m->d_z, m->d_info + p0, k, (real *)m->d_x, m->th, m->sigma2, dbg_skip);
        const int below = m->max_nt - k - 1;
        if (below > 0) {
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
            // grid.x = block rows below + 1 look-ahead workgroup
            hipLaunchKernelGGL((chol_panel_kernel<D, FAM, FUSE>), dim3((unsigned)(below + 1), (unsigned)np), dim3(256), 0, s,
                               m->d_desc + p0, (real *)m->d_a, (real *)m->d_inv, (real *)m->d_y, (real *)m->d_z, ytmp, k, (real *)m->d_x, m->th, m->sigma2);
            if (fine) {
                PMK_HIP(hipEventRecord(c->panel_ev[(size_t)k].second, s));
                c->panel_n = k + 1;
            }
        }
    }
    PMK_HIP(hipGetLastError());
    return 0;
}

// fuse != 0: the kernel matrix is evaluated inside the factorisation kernels (no kmat_slab_kernel pass)
int launch_cholesky(pmk_model *m, hipStream_t s, int64_t p0, int64_t np, int fuse)
{
    if (!fuse) return launch_cholesky_T<1, 0, 0>(m, s, p0, np);
    const bool s34 = m->th.family == PMK_SPLINE34;
    switch (m->D) {
    case 1: return s34 ? launch_cholesky_T<1, PMK_SPLINE34, 1>(m, s, p0, np) : launch_cholesky_T<1, 0, 1>(m, s, p0, np);
    case 2: return s34 ? launch_cholesky_T<2, PMK_SPLINE34, 1>(m, s, p0, np) : launch_cholesky_T<2, 0, 1>(m, s, p0, np);
    case 3: return s34 ? launch_cholesky_T<3, PMK_SPLINE34, 1>(m, s, p0, np) : launch_cholesky_T<3, 0, 1>(m, s, p0, np);
    case 4: return s34 ? launch_cholesky_T<4, PMK_SPLINE34, 1>(m, s, p0, np) : launch_cholesky_T<4, 0, 1>(m, s, p0, np);
    default: set_error("unsupported input dimension %d", m->D); return -2;
    }
}

// -(L[ss])^-1 of every 32 x 32 diagonal block of factors that were loaded from the host (pmk_model_load)
__global__ __launch_bounds__(64) void ninv_from_slab_kernel(const PatchDesc *__restrict__ descs, const real *__restrict__ A,
                                                            real *__restrict__ ninv)
{
    const PatchDesc pd = descs[blockIdx.y];
    const int blk = blockIdx.x;                 // 32-block index along the diagonal
    if (blk >= 4 * pd.nt) return;
    const int c = threadIdx.x;
    if (c >= SB) return;
    const real *Dg = A + pd.aoff + (int64_t)SB * blk + (int64_t)SB * blk * pd.ld;
    real x[SB];
#pragma unroll
    for (int i = 0; i < SB; ++i) {
        real sacc = (i == c) ? 1.0 : 0.0;
#pragma unroll
        for (int l = 0; l < i; ++l) sacc -= Dg[i + (int64_t)l * pd.ld] * x[l];
        x[i] = sacc / Dg[i + (int64_t)i * pd.ld];
    }
    real *out = ninv + pd.ioff + (int64_t)blk * (SB * SB);
#pragma unroll
    for (int i = 0; i < SB; ++i) out[i + SB * c] = (i >= c) ? -x[i] : 0.0;
}

int launch_ninv_from_slabs(pmk_model *m, hipStream_t s)
{
    hipLaunchKernelGGL(ninv_from_slab_kernel, dim3((unsigned)(4 * m->max_nt), (unsigned)m->P), dim3(64), 0, s, m->d_desc,
                       (real *)m->d_a, (real *)m->d_inv);
    PMK_HIP(hipGetLastError());
    return 0;
}

int launch_backsolve(pmk_model *m, hipStream_t s, int64_t p0, int64_t np)
{
    const size_t fixed = sizeof(real) * (TILE + 10 * SB * (SB + 1));
    const size_t csb = sizeof(real) * (size_t)m->max_nt * TILE;
    const int cs_in_lds = fixed + csb <= 150 * 1024;
    const size_t lds = fixed + (cs_in_lds ? csb : 0);
    static bool attr_set = false;
    if (!attr_set) {      // more than the default 64 KB of dynamic LDS
        PMK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(chol_backsolve_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(chol_backsolve_kernel, dim3((unsigned)np), dim3(1024), lds, s, m->d_desc + p0, (real *)m->d_a, (real *)m->d_inv,
                       (real *)m->d_z, (real *)m->d_c, cs_in_lds);
    PMK_HIP(hipGetLastError());
    return 0;
}

}  // namespace PMK_NS
}  // namespace pmk
