// K2/K3: batched blocked Cholesky (left-looking, 128-wide block columns) + the two triangular
// solves for the GP weights.  Replaces, per patch, `cholesky(U)` and `c = U\y` of the reference
// (src/RKHS/mixtureGP.jl:106-112): one factorisation serves both.
//
//   chol_first_kernel : one workgroup per patch: potrf of the first diagonal tile A[0,0] (+ its inverted
//                       32 x 32 diagonal blocks and z_0 = L[00]^-1 y_0).
//   chol_step_kernel  : block column k of every active patch, one launch.  Workgroup = one 128-row block row i > k:
//                           T = A[i,k] - L[i,0:k] L[k,0:k]^T      (MFMA: the contraction the north star prices)
//                           L[i,k] = T L[kk]^-T                    (in-register block substitution on MFMA)
//                       The workgroup of block row k+1 is the CRITICAL one and continues alone:
//                           A[k+1,k+1] -= L[k+1,0:k+1] L[k+1,0:k+1]^T   (look-ahead, MFMA; column k it has just made)
//                           rhs = y_{k+1} - L[k+1,0:k+1] z_{0:k+1}
//                           potrf of the tile in registers -> L[k+1,k+1], -D^-1 blocks, z_{k+1}
//                       so the next launch finds its diagonal block ready: there is no separate (serial,
//                       latency-bound) diagonal launch between two steps, its ~90 us per step hide behind the
//                       other block rows of the same launch.  The critical workgroups get the lowest logical ids
//                       of their XCD, so they are dispatched first.
// Schedule: END-ALIGNED.  Patch p (nt_p tiles) runs its block column k at launch l = k + (max_nt - nt_p): every
// active patch then has exactly max_nt - l - 1 block rows below the diagonal (the grid has no empty workgroups),
// and the long last steps always see every patch -- ragged batches do not end in launches that only the largest
// patches populate.  Patches are visited through `order` (sorted by nt, largest first): the active ones are a prefix.
// The slab is read once per block column (left-looking): reads only, no trailing-matrix
// read-modify-write.  chol_backsolve_kernel then gives c = L^-T z.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <numeric>

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

// LDS of a factorisation workgroup: first the TRSM operands of the block row (TRI_LDS_DOUBLES), later -- in the
// critical workgroup, once the block substitution is done -- the scratch of the tile potrf, carved from the same array
constexpr int POTRF_DBLK = 0;                               // [4][SB][SB+1] the four diagonal 32 x 32 blocks of L[kk]
constexpr int POTRF_COL = POTRF_DBLK + 4 * SB * (SB + 1);   // [TILE+1] scaled pivot column; [TILE] = forward-solve entry
constexpr int POTRF_RHS = POTRF_COL + TILE + 8;             // [2*TILE]
constexpr int POTRF_SDIAG = POTRF_RHS + 2 * TILE;           // [1]
constexpr int POTRF_BAD = POTRF_SDIAG + 2;                  // int
constexpr int POTRF_END = POTRF_BAD + 2;
constexpr int PARTIAL_TILE = TILE * TILE / 2;      // a 128 x 128 partial product tile in real2 units
static_assert(POTRF_END <= TRI_LDS_DOUBLES, "potrf scratch must fit into the TRSM operand array");

// ---------------------------------------------------------------------------------------------
// Stores of the task-queue path (chol_queue_kernel): a tile that another workgroup of the SAME launch will read is
// written through (sc1: the bytes leave the XCD's L2 at once), so the hand-off needs no release fence -- every storing
// wave drains its stores (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, one lane sets the flag, and the consumer
// invalidates its L1 once behind its poll (MI355X_MICROARCH.md, inter-workgroup visibility, form R1).  The asm stores
// are invisible to the compiler's wait counting, which only makes its own counted waits stricter.
// ---------------------------------------------------------------------------------------------
template <bool SC1>
__device__ __forceinline__ void put2(real *p, real2_t v)
{
    if (SC1) {
#ifdef PMK_REAL_F32
        asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#else
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#endif
    } else {
        *reinterpret_cast<real2_t *>(p) = v;
    }
}
template <bool SC1>
__device__ __forceinline__ void put1(real *p, real v)
{
    if (SC1) {
#ifdef PMK_REAL_F32
        asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#else
        asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#endif
    } else {
        *p = v;
    }
}
// a load the compiler keeps on the vector path (and re-issues every time)
__device__ __forceinline__ real load_volatile(const real *p) { return *reinterpret_cast<const volatile real *>(p); }
// all stores of this wave have left (the asm stores above included)
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// Entered by all threads of the workgroup: every wave's stores drained, then this CU's L1 dropped, so that plain loads
// behind it see what the workgroup itself -- or, behind a matched poll, another workgroup -- has written through.
__device__ __forceinline__ void workgroup_refresh()
{
    drain_stores();
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// potrf of one 128 x 128 diagonal tile (already updated, lower part in the slab at Akk), right-looking, the tile
// distributed over the registers of all 256 threads: thread (tr, tc) of a 16 x 16 grid holds
// A[tr + 16 a][tc + 16 b], a, b < 8; the threads with tr == 0 also carry the right-hand side (lds[POTRF_RHS + 0..127],
// written by the caller before a barrier) as an extra row, which the elimination turns into z = L[kk]^-1 rhs.
// Per pivot: the 16 threads that own column j scale it and publish it through LDS, one barrier, every thread applies
// the rank-1 update to its own entries, one barrier.  The pivot's reciprocal square root comes from v_rsq + Newton
// steps: the 128 pivots are a serial latency chain, this is its critical path.
// Writes L[kk] (strict upper part of the slab block zeroed), the negated inverses of its four 32 x 32 diagonal
// blocks (operands of every later block substitution), z, and info (first non-positive pivot, 1-based, once).
// Must be entered by all 256 threads; the caller has synchronised after its last write to Akk and rhs.
// ---------------------------------------------------------------------------------------------
template <bool SC1 = false>
__device__ __forceinline__ void tile_potrf(real *__restrict__ Akk, int64_t ld, real *lds,
                                           real *__restrict__ ninv_k, real *__restrict__ z_k,
                                           int32_t *__restrict__ info_p, int k)
{
#define DBLK(b, i, j) lds[POTRF_DBLK + ((b) * SB + (i)) * (SB + 1) + (j)]
#define COL(i) lds[POTRF_COL + (i)]
#define SDIAG lds[POTRF_SDIAG]
#define SBAD lds[POTRF_BAD]          /* first failed pivot (1-based) as a real: exact up to 2^24 */
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));       // opaque per call: nothing lane-dependent is hoisted out of a caller's task loop
    const int lane = tid & 63, wave = tid >> 6;
    const int tr = tid & 15, tc = tid >> 4;
    real a_[8][8], rr[8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 8; ++b2)
            a_[a][b2] = (b2 <= a) ? Akk[(tr + 16 * a) + (int64_t)(tc + 16 * b2) * ld] : (real)0;
#pragma unroll
    for (int b2 = 0; b2 < 8; ++b2) rr[b2] = lds[POTRF_RHS + tc + 16 * b2];
    if (tid == 0) { SDIAG = a_[0][0]; SBAD = 0; }
    __syncthreads();
    // two-level pivot loop: the 16-column block index bj is a compile-time constant in each copy of the body, so
    // a_[.][bj] is a static register reference and the update loops run over exactly the live blocks -- no
    // per-pivot scalar branches (an earlier version guarded every 16 x 16 block with a runtime test: ~40
    // s_cbranch per pivot cost more than the arithmetic they skipped)
#pragma unroll
    for (int bj = 0; bj < 8; ++bj) {
#pragma clang loop unroll(disable)      // 128 unrolled pivot bodies cross-schedule into hundreds of spilled registers
        for (int jj = 0; jj < 16; ++jj) {
            const int j = 16 * bj + jj;
            if (tc == jj) {
                real d = SDIAG;
                if (!(d > (real)0)) {                     // not positive definite (or NaN): record, keep going
                    if (tr == 0 && SBAD == (real)0) SBAD = (real)(k * TILE + j + 1);
                    d = 1;
                }
                const real rs = rsqrt_real(d);
                const real ljj = d * rs;
#pragma unroll
                for (int a = bj; a < 8; ++a) {
                    const int r = tr + 16 * a;
                    const real v = (r > j) ? a_[a][bj] * rs : ((r == j) ? ljj : a_[a][bj]);
                    a_[a][bj] = v;
                    COL(r) = (r > j) ? v : (real)0;
                }
                if (tr == 0) {                            // the extra row: z_j = rhs_j / L[j][j]
                    rr[bj] = rr[bj] * rs;
                    COL(TILE) = rr[bj];
                }
            }
            __syncthreads();
            real cr[8], cc[8];
#pragma unroll
            for (int a = bj; a < 8; ++a) { cr[a] = COL(tr + 16 * a); cc[a] = COL(tc + 16 * a); }
#pragma unroll
            for (int a = bj; a < 8; ++a)
#pragma unroll
                for (int b2 = bj; b2 <= a; ++b2) a_[a][b2] -= cr[a] * cc[b2];   // col[] is 0 for rows <= j
            if (tr == 0) {
                const real zj = COL(TILE);
#pragma unroll
                for (int b2 = bj; b2 < 8; ++b2) rr[b2] -= zj * cc[b2];
            }
            // publish the next pivot's diagonal entry: A[j+1][j+1] lives in block (bj, bj), or in block
            // (bj+1, bj+1) of thread (0, 0) when j+1 starts the next block
            if (jj < 15) {
                if (tr == jj + 1 && tc == jj + 1) SDIAG = a_[bj][bj];
            } else if (bj < 7) {
                if (tid == 0) SDIAG = a_[bj + 1 < 8 ? bj + 1 : 7][bj + 1 < 8 ? bj + 1 : 7];
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
            put1<SC1>(Akk + r + (int64_t)c * ld, v);
            if ((r >> 5) == (c >> 5)) DBLK(r >> 5, r & 31, c & 31) = v;
        }
    if (tr == 0) {
#pragma unroll
        for (int b2 = 0; b2 < 8; ++b2) put1<SC1>(z_k + tc + 16 * b2, rr[b2]);
    }
    __syncthreads();
    if (tid == 0 && SBAD != (real)0 && *info_p == 0) *info_p = (int32_t)SBAD;
    // ---- negated inverses of the four 32 x 32 diagonal blocks: wave w inverts block w, one thread per column
    if (lane < SB) {
        const int c = lane;
        real xcol[SB];
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            real sacc = (i == c) ? (real)1 : (real)0;
#pragma unroll
            for (int l = 0; l < i; ++l) sacc -= DBLK(wave, i, l) * xcol[l];
            xcol[i] = sacc / DBLK(wave, i, i);
        }
        real *Ni = ninv_k + (int64_t)wave * (SB * SB);
#pragma unroll
        for (int i = 0; i < SB; ++i) put1<SC1>(Ni + i + SB * c, (i >= c) ? -xcol[i] : (real)0);
    }
#undef DBLK
#undef COL
#undef SDIAG
#undef SBAD
}

#ifdef PMK_TRACE
// diagnostic build only (make variant VFLAGS=-DPMK_TRACE=<launch>): wall-clock stamps (100 MHz) of every workgroup of
// ONE step launch, read back by tools/step_trace.py through pmk_trace_dump; never compiled into libpmk_hip.so
constexpr int TRACE_WORDS = 8, TRACE_MAX_WG = 8192;
__device__ unsigned long long g_trace[TRACE_WORDS * TRACE_MAX_WG];
#define PMK_STAMP(i)                                                                                    \
    do {                                                                                                \
        if (launch == PMK_TRACE && threadIdx.x == 0 && blockIdx.x < TRACE_MAX_WG)                       \
            g_trace[TRACE_WORDS * blockIdx.x + (i)] = __builtin_amdgcn_s_memrealtime();                 \
    } while (0)
#else
#define PMK_STAMP(i)
#endif

// first diagonal tile of every patch: nothing to apply, rhs = y_0
__global__ __launch_bounds__(256, 2) void chol_first_kernel(const PatchDesc *__restrict__ descs, real *__restrict__ A,
                                                            real *__restrict__ ninv, const real *__restrict__ y,
                                                            real *__restrict__ z, int32_t *__restrict__ info,
                                                            unsigned long long *__restrict__ clk)
{
    __shared__ real lds[POTRF_END];
    const PatchDesc pd = descs[blockIdx.x];
    // the first launch of a factorisation also clears what the later ones accumulate into (nine small memsets otherwise):
    // the patch's status word and the step launches' shader-clock stamps (the strip kernel's pair, words 128-129 of each
    // XCD's block, stays)
    if (threadIdx.x == 0) info[blockIdx.x] = 0;
    if (clk && blockIdx.x == 0)
        for (int e = threadIdx.x; e < 8 * 128; e += 256) clk[130 * (e >> 7) + (e & 127)] = 0;
    if (threadIdx.x < TILE) lds[POTRF_RHS + threadIdx.x] = y[pd.yoff + threadIdx.x];
    __syncthreads();
    tile_potrf(A + pd.aoff, pd.ld, lds, ninv + pd.ioff, z + pd.yoff, info + blockIdx.x, 0);
}

// ---------------------------------------------------------------------------------------------
// The two building blocks of a factorisation step, for one workgroup of 256 threads.
// ---------------------------------------------------------------------------------------------
// Block row `row` of block column k of one patch:  T = A[row,k] - L[row,0:k] L[k,0:k]^T  on MFMA (4 waves x (32 rows x
// 128 columns), the waves independent after the operand staging), then  L[row,k] = T L[kk]^-T  by in-register block
// substitution.  SPLIT: the deep product arrives as `nsplit` partial tiles instead (chol_partial_kernel).
// Entered by all threads; contains one barrier (after the staging of the TRSM operands in `lds`); the caller
// synchronises before `lds` is reused.
// KD > 0 (fused kernel-matrix build, Spline34 in KD dimensions): the tile A[row, k] has never been written -- its
// entries are evaluated here, at their one and only use, from the patch's coordinates (xs, SoA) with the same
// kern_eval as K1: the same bits, without the write of the tile by K1 and its read here (the strictly lower 128 x 128
// tiles are 7/8 of a 2000-point patch's kernel matrix).  Padding rows carry coordinates of 1e300, where the compact
// profile is exactly 0 (a padded row never meets a padded column below the diagonal tiles).
template <int SPLIT, bool SC1 = false, int KD = 0>
__device__ __forceinline__ void block_row_update(const PatchDesc &pd, real *__restrict__ S, const real *__restrict__ ninv_p,
                                                 int k, int row, real *lds, const real2_t *__restrict__ pt, int nsplit,
                                                 const real *__restrict__ xs = nullptr, const pmk_kernel_desc *th = nullptr)
{
    // the thread index is made opaque per call: called from inside a loop, the compiler otherwise hoists the 32
    // lane-dependent tile offsets (and the 64-bit addresses built on them) out of the loop, keeps them live across the
    // GEMM and spills them
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int64_t ld = pd.ld;
    const int64_t c0 = (int64_t)k * TILE;
    const int64_t r0 = (int64_t)row * TILE + 32 * wave;
    // a wave whose 32 rows all lie in the identity padding (row index >= n) has nothing to compute: those rows of
    // L stay zero.  It still takes part in the operand staging and its barrier.  (n = 2000: one wave of the last
    // block row in every step, 4 % of the MFMA work of a fit.)
    const bool live = r0 < pd.n;
    real *out = S + r0 + 2 * (lane & 15) + c0 * ld;   // rows of this lane, first column of the block column
    // acc starts as -A[rows, block column k]; the GEMM adds L[rows,0:k] L[k,0:k]^T, so acc = -T.  The tile comes
    // from HBM (first touch): its loads are issued before the operand staging so that the two latencies overlap
    // instead of adding up.
    WaveTile<4, 1> acc;   // I = the 128 columns of block column k, J = 32 rows
    if (KD == 0 && live) {
#pragma unroll
        for (int fi = 0; fi < 8; ++fi)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cl = tile_i(fi, lane, q);
                const real2_t a = *reinterpret_cast<const real2_t *>(out + cl * ld);
                acc.f[fi][0][q] = a[0];
                acc.f[fi][1][q] = a[1];
            }
    }
    __builtin_amdgcn_sched_barrier(0);      // keep the tile loads ahead of the staging loads
    // the TRSM operands (off-diagonal 32-blocks of L[kk] and the four -D^-1 blocks) go to LDS once per workgroup
    stage_tri_operands(lds, S + c0 + c0 * ld, ld, ninv_p + (int64_t)k * (4 * SB * SB), tid, 256);
    if (KD > 0 && live) {
        // K[rows, block column k]: the lane's two rows against its 32 columns, under the latency of the staging loads
        constexpr int DD = KD > 0 ? KD : 1;
        real xr[2][DD];
#pragma unroll
        for (int d = 0; d < DD; ++d) {
            const real2_t v = *reinterpret_cast<const real2_t *>(xs + (int64_t)d * ld + r0 + 2 * (lane & 15));
            xr[0][d] = v[0];
            xr[1][d] = v[1];
        }
#pragma unroll
        for (int pi = 0; pi < 4; ++pi)
#pragma unroll
            for (int q = 0; q < 4; ++q) {          // static accumulator indices: fully unrolled (64 evaluations)
                real xc[2][DD];
#pragma unroll
                for (int d = 0; d < DD; ++d) {
                    const real2_t v = *reinterpret_cast<const real2_t *>(xs + (int64_t)d * ld + c0 + 32 * pi + 2 * frag_irow(lane >> 4, q));
                    xc[0][d] = v[0];
                    xc[1][d] = v[1];
                }
#pragma unroll
                for (int ei = 0; ei < 2; ++ei)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        acc.f[2 * pi + ei][e][q] = kern_eval<DD, PMK_SPLINE34, real>(*th, xr[e], xc[ei]);
            }
    }
    __syncthreads();
    if (!live) {
        if (KD > 0) {
            // nobody has written these rows of the slab (K1 left the tiles below the diagonal alone): the identity padding's
            // zeros, which the look-ahead of the last diagonal tile reads, are put down here
#pragma unroll
            for (int fi = 0; fi < 8; ++fi)
#pragma unroll
                for (int q = 0; q < 4; ++q) put2<SC1>(out + tile_i(fi, lane, q) * ld, real2_t{0, 0});
        }
        return;
    }
#pragma unroll
    for (int fi = 0; fi < 8; ++fi)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc.f[fi][0][q] = -acc.f[fi][0][q];
            acc.f[fi][1][q] = -acc.f[fi][1][q];
        }
    if (SPLIT) {
        pt += wave * (PARTIAL_TILE / 4) + lane;
        for (int sp = 0; sp < nsplit; ++sp, pt += PARTIAL_TILE)
#pragma unroll
            for (int fi = 0; fi < 8; ++fi)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const real2_t v = pt[(fi * 4 + q) * 64];
                    acc.f[fi][0][q] += v[0];
                    acc.f[fi][1][q] += v[1];
                }
    } else if (k > 0) {
        gemm_nt<4, 1, PF_CHOL, PFJ_CHOL>(acc, S + c0, ld, S + r0, ld, k * TILE, lane);
    }
    // L[rows, k]^T = L[kk]^-1 T^T = -L[kk]^-1 (-T)^T : exactly what the block substitution returns
    tri_solve_inplace<1>(acc, lds, lane);
#pragma unroll
    for (int fi = 0; fi < 8; ++fi)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cl = tile_i(fi, lane, q);
            real2_t o;
            o[0] = acc.f[fi][0][q];
            o[1] = acc.f[fi][1][q];
            put2<SC1>(out + cl * ld, o);
        }
}

// The continuation of the workgroup that made block row k + 1 of block column k: look-ahead
//     A[k+1,k+1] -= L[k+1,0:k+1] L[k+1,0:k+1]^T,    rhs = y_{k+1} - L[k+1,0:k+1] z_{0:k+1},
// then the potrf of the tile (-> L[k+1,k+1], its -D^-1 blocks, z_{k+1}).  Entered after a barrier that follows the
// stores of L[k+1, k] (same CU: visible through its L1) and the last use of `lds`.
// Generalised to a range of block columns [cb, ce) of block row t = k + 1 (the task queue applies columns 0 .. k - 1 as
// soon as they are final and column k inside the potrf task): rhs_src = the right-hand side so far (y_t, or what an
// earlier range left), rhs_dst = where this range leaves it (nullptr: in LDS, for the potrf that follows).  The
// right-hand side is reduced block column by block column, so that any split of the range gives the same bits.
template <int SPLIT, bool SC1 = false>
__device__ __forceinline__ void lookahead_update(const PatchDesc &pd, real *__restrict__ S,
                                                 const real *rhs_src, const real *z_p, int k, real *lds,
                                                 const real2_t *__restrict__ pt, int nsplit, int cb, int ce,
                                                 real *rhs_dst = nullptr)
{
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                     // see block_row_update
    const int lane = tid & 63, wave = tid >> 6;
    const int64_t ld = pd.ld;
    const int64_t c0 = (int64_t)k * TILE, t0 = c0 + TILE;
    const int K2 = (ce - cb) * TILE;                  // depth of this range
    const int64_t cs = (int64_t)cb * TILE;            // its first column
    const int h = wave >> 1, g = wave & 1;            // 64-row half, 64-column half of the tile
    real *Att = S + t0 + t0 * ld;
    if (SPLIT) {
        // the look-ahead tile arrives as partial products too (block columns 0..k-1); block column k -- made just
        // before -- is applied here.  All four waves, 32 rows x 128 columns each: the part above the diagonal is
        // computed along (never read by anyone: the potrf below takes the lower triangle only).
        WaveTile<4, 1> acc;
        real *outl = Att + 32 * wave + 2 * (lane & 15);
#pragma unroll
        for (int fi = 0; fi < 8; ++fi)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const real2_t a = *reinterpret_cast<const real2_t *>(outl + tile_i(fi, lane, q) * ld);
                acc.f[fi][0][q] = -a[0];
                acc.f[fi][1][q] = -a[1];
            }
        pt += wave * (PARTIAL_TILE / 4) + lane;
        for (int sp = 0; sp < nsplit; ++sp, pt += PARTIAL_TILE)
#pragma unroll
            for (int fi = 0; fi < 8; ++fi)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const real2_t v = pt[(fi * 4 + q) * 64];
                    acc.f[fi][0][q] += v[0];
                    acc.f[fi][1][q] += v[1];
                }
        if (ce > cb) gemm_nt<4, 1, PF_CHOL, PFJ_CHOL>(acc, S + t0 + c0 * ld, ld, S + t0 + 32 * wave + c0 * ld, ld, TILE, lane);
#pragma unroll
        for (int fi = 0; fi < 8; ++fi)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                real2_t o;
                o[0] = -acc.f[fi][0][q];
                o[1] = -acc.f[fi][1][q];
                *reinterpret_cast<real2_t *>(outl + tile_i(fi, lane, q) * ld) = o;
            }
        if (tid < TILE && ce > cb) lds[POTRF_RHS + tid] = (real)0;       // z comes from the solve sweeps of the split path
    } else if (!(h == 0 && g == 1)) {
        // lower 64 x 64 sub-tiles: acc = -A up-front (all the sub-tile's loads in flight at once), the GEMM adds
        // L L^T, the store negates
        WaveTile<2, 2> acc;
#pragma unroll
        for (int fi = 0; fi < 4; ++fi)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int pj = 0; pj < 2; ++pj) {
                    const int cl = 64 * g + tile_i(fi, lane, q);
                    const int rl = 64 * h + 32 * pj + 2 * (lane & 15);
                    const real2_t a = *reinterpret_cast<const real2_t *>(Att + rl + (int64_t)cl * ld);
                    acc.f[fi][2 * pj][q] = -a[0];
                    acc.f[fi][2 * pj + 1][q] = -a[1];
                }
        if (K2 > 0) gemm_nt<2, 2, PF_DIAG>(acc, S + t0 + 64 * g + cs * ld, ld, S + t0 + 64 * h + cs * ld, ld, K2, lane);
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
                    put2<SC1>(Att + rl + (int64_t)cl * ld, a);
                }
    } else {
        // the wave without a GEMM sub-tile does the forward-solve right-hand side, two rows per lane
        const real *Lr = S + t0 + 2 * lane;
        real2_t r = *reinterpret_cast<const real2_t *>(rhs_src + 2 * lane);
        for (int jb = cb; jb < ce; ++jb) {
            real2_t s0 = {0.0, 0.0}, s1 = {0.0, 0.0}, s2 = {0.0, 0.0}, s3 = {0.0, 0.0};
            for (int c = jb * TILE; c < (jb + 1) * TILE; c += 16) {     // 16 independent column loads in flight per batch
                real2_t av[16];
                real zv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) av[u] = *reinterpret_cast<const real2_t *>(Lr + (int64_t)(c + u) * ld);
                // z comes from other workgroups of the same launch in the queue form: vector loads (a uniform index on
                // a const restrict pointer would go through the scalar cache, which no acquire refreshes)
#pragma unroll
                for (int u = 0; u < 16; ++u) zv[u] = SC1 ? load_volatile(z_p + c + u) : z_p[c + u];
#pragma unroll
                for (int u = 0; u < 16; u += 4) {
                    s0 += av[u] * zv[u]; s1 += av[u + 1] * zv[u + 1];
                    s2 += av[u + 2] * zv[u + 2]; s3 += av[u + 3] * zv[u + 3];
                }
            }
            r -= (s0 + s1) + (s2 + s3);
        }
        if (rhs_dst) put2<SC1>(rhs_dst + 2 * lane, r);
        else *reinterpret_cast<real2_t *>(lds + POTRF_RHS + 2 * lane) = r;
    }
}

template <int SPLIT>
__device__ __forceinline__ void lookahead_potrf(const PatchDesc &pd, real *__restrict__ S, real *__restrict__ ninv_p,
                                                const real *__restrict__ y_p, real *__restrict__ z_p,
                                                int32_t *__restrict__ info_p, int k, real *lds,
                                                const real2_t *__restrict__ pt, int nsplit)
{
    lookahead_update<SPLIT, false>(pd, S, y_p + (int64_t)(k + 1) * TILE, z_p, k, lds, pt, nsplit, 0, k + 1);
    __syncthreads();
    const int64_t t0 = (int64_t)(k + 1) * TILE;
    tile_potrf<false>(S + t0 + t0 * pd.ld, pd.ld, lds, ninv_p + (int64_t)(k + 1) * (4 * SB * SB), z_p + t0, info_p, k + 1);
}

// (patch slot, block row) of a step-type launch from the hardware block id.  Blocks are dealt round-robin over the 8
// XCDs (block b runs on XCD b % 8): XCD x takes the patch slots x, x + 8, x + 16, ... (sizes interleaved, so ragged
// batches load the XCDs evenly) and ALL block rows of those patches -- they stream the same block row k of L, which
// that XCD's L2 then fetches once.  Within an XCD the critical workgroups (bx == 0) have the lowest ids: they are
// dispatched first.  The grid is padded to 8 x ceil(nslots / 8) x G; surplus workgroups get slot = -1.
__device__ __forceinline__ void step_slot(int nslots, int G, int &slot, int &bx)
{
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int per = (nslots - xcd + 7) >> 3;             // patch slots of this XCD
    slot = -1; bx = 0;
    if (idx >= per * G) return;
    int loc;
    if (idx < per) { loc = idx; bx = 0; }
    else { const int j = idx - per; loc = j / (G - 1); bx = 1 + j % (G - 1); }
    slot = xcd + 8 * loc;
}

// ---------------------------------------------------------------------------------------------
// one block column of every active patch.  Workgroup = one 128-row block row; two workgroups share a CU (2 waves per
// SIMD).  The workgroup of block row k + 1 (bx == 0) is the critical one and continues alone (lookahead_potrf).
// SPLIT > 0 (single large problems, P too small to fill the chip with one workgroup per block row): the deep GEMMs
// were done by chol_partial_kernel, SPLIT-way split along K, and left as partial tiles; this kernel then adds them up,
// applies block column k itself (128 deep) and carries on as above -- except that the forward-solve right-hand side is
// not carried along (z comes from the separate solve sweeps of that path).
// ---------------------------------------------------------------------------------------------
template <int SPLIT, int KD = 0>
__global__ __launch_bounds__(256, 2) void chol_step_kernel(const PatchDesc *__restrict__ descs,
                                                           const int32_t *__restrict__ order, int nactive, int G,
                                                           int launch, int max_nt, real *__restrict__ A,
                                                           real *__restrict__ ninv, const real *__restrict__ y,
                                                           real *__restrict__ z, int32_t *__restrict__ info,
                                                           const real2_t *__restrict__ partial, int nsplit,
                                                           unsigned long long *__restrict__ clk,
                                                           const real *__restrict__ x, pmk_kernel_desc th,
                                                           int nsplit_look)
{
    __shared__ real lds[TRI_LDS_DOUBLES];
    int slot, bx;
    // shader-clock probe (workgroups 0..7 are critical ones, one per XCD: they live through more than half of the launch)
    const bool probe = clk && blockIdx.x < 8 && threadIdx.x == 0 && launch < 64;
    unsigned long long c0 = 0, r0 = 0;
    if (probe) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    if (SPLIT) {
        // few patches: their block rows are dealt over ALL XCDs (one XCD per patch would leave most of the chip idle);
        // critical workgroups first
        const int lid = blockIdx.x;
        if (lid >= nactive * (G + 1)) return;
        if (lid >= nactive * G) {
            if (nsplit_look == 0) return;                                // PMK_SPLIT_PREREDUCE=0: the factorising workgroup does it
            // one more workgroup per patch: the partial tiles of the NEXT diagonal tile (block columns 0..k-1) are folded
            // into the slab here, beside the block rows, instead of by the workgroup that factorises it (whose chain of
            // partial sums + block column k + potrf is what a split step waits for).  Same sums in the same order.
            slot = lid - nactive * G;
            const int pid = order[slot];
            const PatchDesc pd = descs[pid];
            const int k = launch - (max_nt - pd.nt);
            lookahead_update<1, false>(pd, A + pd.aoff, nullptr, nullptr, k, lds,
                                       partial + (((int64_t)slot * (G + 1) + G) * nsplit) * PARTIAL_TILE, nsplit_look, 0, 0);
            return;
        }
        if (lid < nactive) { slot = lid; bx = 0; }
        else { const int j = lid - nactive; slot = j / (G - 1); bx = 1 + j % (G - 1); }
    } else {
        step_slot(nactive, G, slot, bx);
        if (slot < 0) return;
    }
    const int pid = order[slot];
    const PatchDesc pd = descs[pid];
    const int k = launch - (max_nt - pd.nt);      // this patch's block column (end-aligned schedule): 0 <= k < nt - 1
    real *S = A + pd.aoff;
#ifdef PMK_TRACE
    const int tid = threadIdx.x, lane = tid & 63;
    if (launch == PMK_TRACE && tid == 0 && blockIdx.x < TRACE_MAX_WG) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        g_trace[TRACE_WORDS * blockIdx.x + 0] = ((unsigned long long)slot << 32) | (unsigned)bx;
        g_trace[TRACE_WORDS * blockIdx.x + 1] = ((unsigned long long)hwid << 32) | (xcc & 0xf);
        g_trace[TRACE_WORDS * blockIdx.x + 6] = 0;
        g_trace[TRACE_WORDS * blockIdx.x + 7] = __builtin_amdgcn_s_memtime();       // shader clock at the start
    }
#endif
    PMK_STAMP(2);
    block_row_update<SPLIT, false, KD>(pd, S, ninv + pd.ioff, k, k + 1 + bx, lds,
                                       SPLIT ? partial + (((int64_t)slot * (G + 1) + bx) * nsplit) * PARTIAL_TILE : nullptr, nsplit,
                                       KD ? x + pd.xoff : nullptr, &th);
    PMK_STAMP(3);
#ifdef PMK_TRACE
    if (launch == PMK_TRACE && tid == 0 && blockIdx.x < TRACE_MAX_WG)      // shader cycles of wave 0's block-row phase
        g_trace[TRACE_WORDS * blockIdx.x + 7] = __builtin_amdgcn_s_memtime() - g_trace[TRACE_WORDS * blockIdx.x + 7];
    if (launch == PMK_TRACE && lane == 0 && blockIdx.x < TRACE_MAX_WG)
        atomicMax(&g_trace[TRACE_WORDS * blockIdx.x + 6], (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
    if (bx != 0 || SPLIT) return;      // split steps: the next diagonal tile is factorised beside the next step's products
    // ================= critical workgroup: look-ahead + potrf of diagonal tile k + 1 =================
    __syncthreads();
    PMK_STAMP(4);
    lookahead_potrf<SPLIT>(pd, S, ninv + pd.ioff, y + pd.yoff, z + pd.yoff, info + pid, k, lds,
                           SPLIT ? partial + (((int64_t)slot * (G + 1) + G) * nsplit) * PARTIAL_TILE : nullptr, nsplit);
    PMK_STAMP(5);
    if (probe) {
        clk[130 * blockIdx.x + 2 * launch] = __builtin_amdgcn_s_memtime() - c0;
        clk[130 * blockIdx.x + 2 * launch + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

// ---------------------------------------------------------------------------------------------
// Task-queue form of the same factorisation: ONE launch for all block columns (opt-in: PMK_CHOL_QUEUE=1; measured
// SLOWER than the step launches at config C, profiles/r03_fit_experiments.txt -- kept as the experiment it is, with its
// timeline tool tools/queue_trace.py).
//
// The step launches above are global barriers: every launch ends with half-empty CUs (the last workgroups of a
// launch finish ~100 us apart) and starts cold, 15 times per fit at config C.  Here the work of all launches is one
// list of tasks per XCD -- POT (potrf of a diagonal tile, after the last block column of its block row has been
// applied to it), ROW (one block row of one block column) and LOOK (the earlier block columns applied to the next
// diagonal tile and its right-hand side as soon as they are final) -- and a resident grid of workgroups (two per CU)
// pulls tasks from the list of the XCD it runs on (HW_REG_XCC_ID) with one returning atomic add.  Dependencies are
// per-patch words in global memory:
//     flag[0]           = number of diagonal tiles factorised (potrf k done, -D^-1 blocks and z_k out  =>  k + 1)
//     flag[1 + i]       = number of block columns of block row i that are final
//     flag[fstride - 1] = the diagonal tile whose look-ahead (block columns 0 .. t - 2) is done
// The list is in an order in which everything a task waits for comes EARLIER in the same list, and a task is only
// held by a running workgroup, so the earliest unfinished task can always proceed: no assumption about dispatch order
// or residency is needed for progress, and none about placement for correctness (every hand-off is write-through
// stores -> drained -> barrier -> flag, poll -> L1 invalidate -> barrier -> loads).  The lists are per XCD because the
// block rows of a patch stream the same block row k of L: one L2 then fetches it once -- a speed choice; a workgroup
// only serves the list of its own XCD, and the host checks after the launch that every list was drained (an XCD
// without workgroups would leave its list untouched).  Spins are bounded (QUEUE_SPIN_TICKS of the 100 MHz clock); a
// time-out sets the error word, every workgroup then stops pulling and the host reports the fit as failed.
// The loop keeps its barriers out of divergent control flow: the scalar work (flags, dequeue, polls) sits in a region
// of wave 0 that is closed before the barrier that follows it (with the flags raised at the loop TAIL the compiler
// merged tail and head regions and moved a barrier into a divergent loop: stale control words, wild task indices).
// ---------------------------------------------------------------------------------------------
struct CholTask {
    int32_t pid;        // patch
    int16_t k;          // block column
    int16_t row;        // block row (ROW, CRIT: k + 1)
    int32_t type;       // QT_*
    int32_t pad_;
};
// POT: potrf of diagonal tile k (k = 0: as K1 left it; k >= 1: block column k - 1 of its block row is applied first);
// ROW: block row `row` of block column k; LOOK: block columns 0 .. k - 2 applied to diagonal tile k and to its
// right-hand side (k >= 2), as soon as block row k is final that far
constexpr int QT_POT = 0, QT_ROW = 1, QT_LOOK = 2;
struct QueueOffsets { int32_t off[9]; };                 // list of XCD x = tasks[off[x] .. off[x + 1])
constexpr int QUEUE_MAX_SEGS = 16;      // the task lists may be cut into several launches (PMK_QUEUE_SEGS, experiments)
constexpr int SCHED_ERR = 0, SCHED_HEADS = 16, SCHED_FLAGS = 16 + 8 * QUEUE_MAX_SEGS;      // int32 words of the scheduling block
constexpr unsigned long long QUEUE_SPIN_TICKS = 300000000ull;       // 3 s

#ifdef PMK_QTRACE
// diagnostic build only (make variant VFLAGS=-DPMK_QTRACE): per task {start, deps met, end} in 100 MHz ticks + where
constexpr int QTRACE_WORDS = 4, QTRACE_MAX = 1 << 18;
__device__ unsigned long long g_qtrace[QTRACE_WORDS * QTRACE_MAX];
#define PMK_QSTAMP(i)                                                                                       \
    do {                                                                                                    \
        if (threadIdx.x == 0 && gtask < QTRACE_MAX) g_qtrace[QTRACE_WORDS * gtask + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define PMK_QSTAMP(i)
#endif

// thread 0 only: wait until *p >= need; false on timeout or when another workgroup has raised the error word
__device__ __forceinline__ bool spin_until_ge(const int32_t *p, int need, const int32_t *err)
{
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) return true;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        __builtin_amdgcn_s_sleep(16);
        if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) return true;
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
        if (__builtin_amdgcn_s_memrealtime() - t0 > QUEUE_SPIN_TICKS) return false;
    }
}

// the flags a task waits for before it starts (fl = the patch's flag words, look = index of its look-ahead word):
//   ROW (k, row): block row k and block row `row` final through column k - 1 (the diagonal tile k is waited for later)
//   LOOK (t)    : block row t final through column t - 2, and LOOK (t - 1) done (one word per patch counts them: they
//                 must finish in order)
//   POT (t >= 1): block row t final through column t - 1, and LOOK (t) done (t >= 2)
__device__ __forceinline__ bool task_ready(const CholTask &c, const int32_t *fl, int look)
{
    auto ge = [&](int i, int v) { return __hip_atomic_load(fl + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= v; };
    if (c.type == QT_ROW) return c.k == 0 || (ge(1 + c.k, c.k) && ge(1 + c.row, c.k));
    if (c.type == QT_LOOK) return ge(1 + c.k, c.k - 1) && (c.k < 3 || ge(look, c.k - 1));
    return c.k == 0 || (ge(1 + c.k, c.k) && (c.k < 2 || ge(look, c.k)));
}
__device__ __forceinline__ bool task_wait(const CholTask &c, const int32_t *fl, int look, const int32_t *err)
{
    if (c.type == QT_ROW) return c.k == 0 || (spin_until_ge(fl + 1 + c.k, c.k, err) && spin_until_ge(fl + 1 + c.row, c.k, err));
    if (c.type == QT_LOOK) return spin_until_ge(fl + 1 + c.k, c.k - 1, err) && (c.k < 3 || spin_until_ge(fl + look, c.k - 1, err));
    return c.k == 0 || (spin_until_ge(fl + 1 + c.k, c.k, err) && (c.k < 2 || spin_until_ge(fl + look, c.k, err)));
}

#ifndef PMK_Q_PLAIN
#define PMK_Q_SC1 true
#else
#define PMK_Q_SC1 false
#endif
// block_row_update for the task queue: the deep product only needs block row k of L (and the task's own row) through
// column k - 1, the block substitution needs the factorised diagonal tile k.  If tile k was already there when the task
// started (`diag_ready`, the rule away from the end of a factorisation) its operands are staged first, under the
// latency of the tile loads, exactly as in block_row_update; otherwise the product runs first -- beside the potrf
// that some other workgroup is still busy with -- and wave 0 then waits for the tile (flag[0] >= k + 1), drops this
// CU's L1 once more, and the operands are staged behind the product.  Returns false on a spin time-out (uniform).
__device__ __forceinline__ bool block_row_update_queue(const PatchDesc &pd, real *__restrict__ S,
                                                       const real *__restrict__ ninv_p, int k, int row, real *lds,
                                                       bool diag_ready, const int32_t *fl, const int32_t *err)
{
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int64_t ld = pd.ld;
    const int64_t c0 = (int64_t)k * TILE;
    const int64_t r0 = (int64_t)row * TILE + 32 * wave;
    const bool live = r0 < pd.n;
    real *out = S + r0 + 2 * (lane & 15) + c0 * ld;
    WaveTile<4, 1> acc;
    if (live) {
#pragma unroll
        for (int fi = 0; fi < 8; ++fi)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cl = tile_i(fi, lane, q);
                const real2_t a = *reinterpret_cast<const real2_t *>(out + cl * ld);
                acc.f[fi][0][q] = a[0];
                acc.f[fi][1][q] = a[1];
            }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (diag_ready) stage_tri_operands(lds, S + c0 + c0 * ld, ld, ninv_p + (int64_t)k * (4 * SB * SB), tid, 256);
    if (live) {
#pragma unroll
        for (int fi = 0; fi < 8; ++fi)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc.f[fi][0][q] = -acc.f[fi][0][q];
                acc.f[fi][1][q] = -acc.f[fi][1][q];
            }
        if (k > 0) gemm_nt<4, 1, PF_CHOL, PFJ_CHOL>(acc, S + c0, ld, S + r0, ld, k * TILE, lane);
    }
    bool ok = true;
    if (!diag_ready) {
        int *ctl = reinterpret_cast<int *>(lds);
        if (__builtin_amdgcn_readfirstlane(tid >> 6) == 0) {
            if (tid == 0) {
                const bool got = spin_until_ge(fl, k + 1, err);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                ctl[2] = got ? 1 : 0;
            }
        }
        __syncthreads();
        ok = __builtin_amdgcn_readfirstlane(ctl[2]) != 0;
        __syncthreads();
        if (ok) stage_tri_operands(lds, S + c0 + c0 * ld, ld, ninv_p + (int64_t)k * (4 * SB * SB), tid, 256);
    }
    __syncthreads();
    if (!live || !ok) return ok;
    tri_solve_inplace<1>(acc, lds, lane);
#pragma unroll
    for (int fi = 0; fi < 8; ++fi)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cl = tile_i(fi, lane, q);
            real2_t o;
            o[0] = acc.f[fi][0][q];
            o[1] = acc.f[fi][1][q];
            put2<PMK_Q_SC1>(out + cl * ld, o);
        }
    return true;
}

__global__ __launch_bounds__(256, 2) void chol_queue_kernel(const PatchDesc *__restrict__ descs,
                                                            const CholTask *__restrict__ tasks, QueueOffsets qo,
                                                            int32_t *sched, int seg, int fstride, real *__restrict__ A,
                                                            real *__restrict__ ninv, const real *__restrict__ y,
                                                            real *__restrict__ z, int32_t *__restrict__ info,
                                                            unsigned long long *__restrict__ clk)
{
    // all of the CU's LDS for two workgroups: the two control words of the task loop live in the staging array
    // (consumed into registers, behind a barrier, before a task touches it)
    __shared__ real lds[TRI_LDS_DOUBLES];
    int *s_ctl = reinterpret_cast<int *>(lds);
    const int tid = threadIdx.x;
    // the scalar work of the loop (dequeue, polls, flags) belongs to wave 0: a uniform branch on the wave number and,
    // inside it, a one-lane region that contains no barrier -- the loop itself stays a plain scalar loop
    const bool wave0 = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
    const bool probe = clk && blockIdx.x < 8 && tid == 0;
    unsigned long long c0 = 0, r0 = 0;
    if (probe) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    (void)hwid;
    const int q = (int)(xcc & 7);
    const int qbase = qo.off[q], nq = qo.off[q + 1] - qo.off[q];
    int32_t *head = sched + SCHED_HEADS + 8 * seg + q, *err = sched + SCHED_ERR, *flags = sched + SCHED_FLAGS;
    const int look = fstride - 1;      // index of a patch's look-ahead word among its flags
    CholTask done = {0, 0, 0, -1, 0};     // the task this workgroup has just finished (its flags are raised at the loop head)
    for (;;) {
        if (wave0) {
            if (tid == 0) {
                if (done.type >= 0) {
                    int32_t *fl = flags + (int64_t)done.pid * fstride;
                    if (done.type == QT_ROW) __hip_atomic_store(fl + 1 + done.row, done.k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else if (done.type == QT_LOOK) __hip_atomic_store(fl + look, (int)done.k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else __hip_atomic_store(fl, done.k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                // Taking a task is one returning add on the head of this XCD's list.  The list is in an order in which
                // everything a task waits for comes earlier, so whatever the wait below is for has been taken by a
                // workgroup that is running: progress needs no assumption about dispatch order or residency.
                int got = -1;
                if (nq > 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                    const int t = __hip_atomic_fetch_add(head, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (t < nq) got = qbase + t;
                }
                int diag = 1;
                if (got >= 0) {
                    const CholTask c = tasks[got];
                    const int32_t *fl = flags + (int64_t)c.pid * fstride;
                    if (!task_wait(c, fl, look, err)) {
                        __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        got = -1;
                    } else if (c.type == QT_ROW) {
                        diag = __hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= c.k + 1;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // ONE L1 invalidate behind the matched polls
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                s_ctl[0] = got;
                s_ctl[1] = diag;
            }
        }
        __syncthreads();
        const int gt = __builtin_amdgcn_readfirstlane(s_ctl[0]);
        const int diag_ready = __builtin_amdgcn_readfirstlane(s_ctl[1]);
        __syncthreads();                 // every wave has read the control words: the array is the task's now
        if (gt < 0) break;
        const CholTask tk = tasks[gt];
#ifdef PMK_QTRACE
        const int gtask = gt;
        if (tid == 0 && gtask < QTRACE_MAX)
            g_qtrace[QTRACE_WORDS * gtask + 3] = ((unsigned long long)hwid << 32) | ((unsigned)q << 28) | blockIdx.x;
#endif
        PMK_QSTAMP(0);
        const PatchDesc pd = descs[tk.pid];
        real *S = A + pd.aoff;
        bool ok = true;
        PMK_QSTAMP(1);
#ifndef PMK_Q_NOCOMPUTE
        if (tk.type == QT_ROW) {
            ok = block_row_update_queue(pd, S, ninv + pd.ioff, tk.k, tk.row, lds, diag_ready != 0,
                                        flags + (int64_t)tk.pid * fstride, err);
        } else {
            // LOOK (t): block columns 0 .. t - 2 onto diagonal tile t and its right-hand side; what is left of the right-hand
            // side is parked in z_t (its final value comes from the potrf task).  POT (t >= 1): block column t - 1 (just
            // finished by the ROW task of this block row) onto both, then the potrf; POT (0): the tile as K1 left it.
            const int t = tk.k;
            const int64_t t0 = (int64_t)t * TILE;
            const bool pot = tk.type == QT_POT;
            if (pot && t == 0) {
                if (tid < TILE) lds[POTRF_RHS + tid] = y[pd.yoff + tid];
            } else {
                const bool from_z = pot && t >= 2;
                lookahead_update<0, PMK_Q_SC1>(pd, S, (from_z ? z : const_cast<real *>(y)) + pd.yoff + t0, z + pd.yoff, t - 1, lds,
                                               nullptr, 1, pot ? t - 1 : 0, pot ? t : t - 1, pot ? nullptr : z + pd.yoff + t0);
            }
            if (pot) {
                workgroup_refresh();          // the tile went out write-through: re-read behind an L1 invalidate
                tile_potrf<PMK_Q_SC1>(S + t0 + t0 * pd.ld, pd.ld, lds, ninv + pd.ioff + (int64_t)t * (4 * SB * SB),
                                      z + pd.yoff + t0, info + tk.pid, t);
            }
        }
        if (!ok) {
            if (wave0) {
                if (tid == 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            break;
        }
#endif
        // end of the task: every wave drains its stores and the workgroup meets; the flags go up at the loop head
        drain_stores();
        __syncthreads();
        done = tk;
        PMK_QSTAMP(2);
    }
    if (probe) {
        clk[130 * blockIdx.x + 0] = __builtin_amdgcn_s_memtime() - c0;
        clk[130 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

// ---------------------------------------------------------------------------------------------
// Split-K path (single large problems: SURVEY section 8(f) rank 3).  With few patches the left-looking step has only
// nt - k - 1 workgroups, each with a GEMM of depth 128 k, and a critical workgroup whose look-ahead is as deep: the
// chip idles and the factorisation is chain-bound.  chol_partial_kernel cuts every deep product of a step -- the G
// block rows AND the look-ahead tile -- into `nsplit` chunks along K, one workgroup each (a 2-D tiling of the step:
// tiles x K chunks), and writes the partial tiles; chol_step_kernel<1> sums them.  Tile layout in memory = the
// accumulator layout ([wave][fragment][register][lane] pairs), so stores and loads are 1 KB coalesced per wave.
// ---------------------------------------------------------------------------------------------
// The first `npot` workgroups of the launch do something else: the diagonal tile that the PREVIOUS split step left to be
// factorised (its look-ahead partial tiles are in that step's buffer, its block row is final) -- look-ahead sum, the
// block column just made, potrf -- beside this step's products, which do not depend on it (they read block columns
// < k; the step kernel that follows needs both).  Inside the step kernel that chain (~200 us, one workgroup, the rest
// of the chip idle) was most of a step of a single large problem.
__global__ __launch_bounds__(256, 2) void chol_partial_kernel(const PatchDesc *__restrict__ descs,
                                                              const int32_t *__restrict__ order, int nactive, int G,
                                                              int nsplit, int launch, int max_nt, real *__restrict__ A,
                                                              real2_t *__restrict__ partial, int npot, int pot_G,
                                                              int pot_nsplit, const real2_t *__restrict__ pot_partial,
                                                              real *__restrict__ ninv, const real *__restrict__ y,
                                                              real *__restrict__ z, int32_t *__restrict__ info)
{
    __shared__ real lds[TRI_LDS_DOUBLES];
    if ((int)blockIdx.x < npot) {
        const int slot = blockIdx.x, pid = order[slot];
        const PatchDesc pd = descs[pid];
        const int k = (launch - 1) - (max_nt - pd.nt);           // the block column of the previous step
        lookahead_potrf<1>(pd, A + pd.aoff, ninv + pd.ioff, y + pd.yoff, z + pd.yoff, info + pid, k, lds,
                           pot_partial + (((int64_t)slot * (pot_G + 1) + pot_G) * pot_nsplit) * PARTIAL_TILE, pot_nsplit);
        return;
    }
    // (patch slot, tile, K chunk) straight from the block id: consecutive blocks = the chunks of one tile, then the next
    // tile of the same patch -- spread over all XCDs (this path exists because there are too few patches to fill them)
    const int tiles = G + 1;
    const int idx = (int)blockIdx.x - npot;
    if (idx >= nactive * tiles * nsplit) return;
    const int slot = idx / (tiles * nsplit), rem = idx - slot * (tiles * nsplit);
    const int t = rem / nsplit, sp = rem - t * nsplit;
    const PatchDesc pd = descs[order[slot]];
    const int k = launch - (max_nt - pd.nt);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const real *S = A + pd.aoff;
    const int64_t ld = pd.ld;
    const int chunk = (k + nsplit - 1) / nsplit;              // block columns per chunk
    const int kb = sp * chunk, ke = min(k, kb + chunk);
    const int64_t rI = (int64_t)(t < G ? k : k + 1) * TILE;                      // I rows: block row k, or k + 1 (look-ahead)
    const int64_t rJ = (int64_t)(t < G ? k + 1 + t : k + 1) * TILE + 32 * wave;  // J rows of this wave
    WaveTile<4, 1> acc;
    acc.zero();
    if (ke > kb && rJ < pd.n)
        gemm_nt<4, 1, PF_CHOL, PFJ_CHOL>(acc, S + rI + (int64_t)kb * TILE * ld, ld, S + rJ + (int64_t)kb * TILE * ld, ld,
                                         (ke - kb) * TILE, lane);
    real2_t *out = partial + (((int64_t)slot * tiles + t) * nsplit + sp) * PARTIAL_TILE + wave * (PARTIAL_TILE / 4) + lane;
#pragma unroll
    for (int fi = 0; fi < 8; ++fi)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            real2_t o;
            o[0] = acc.f[fi][0][q];
            o[1] = acc.f[fi][1][q];
            out[(fi * 4 + q) * 64] = o;
        }
}

// The two triangular solves of the split path, block by block, two launches per block:
//   solve_partial_kernel<DIR> : the long matrix-vector product of block k cut into chunks, one workgroup each
//       DIR = +1 (z = L^-1 y):  p[chunk][row] = sum_{c in chunk of [0, 128 k)}     L[128 k + row, c] z[c]
//       DIR = -1 (c = L^-T z):  p[chunk][col] = sum_{i in chunk of [128 (k+1), ld)} L[i, 128 k + col] c[i]
//   solve_block_kernel<DIR>   : v = rhs_k - sum of the chunks (fixed order: deterministic), then the 128 x 128 diagonal
//       solve by block substitution with the staged 32 x 32 blocks (off-diagonal blocks of L[kk], -D^-1 blocks).
// Block index of a patch at launch j: k = j (forward), k = nt - 1 - j (backward); shorter patches sit out.
template <int DIR>
__global__ __launch_bounds__(256) void solve_partial_kernel(const PatchDesc *__restrict__ descs, int j, int nchunk,
                                                            const real *__restrict__ A, const real *__restrict__ vec,
                                                            real *__restrict__ part)
{
    const PatchDesc pd = descs[blockIdx.y];
    if (j >= pd.nt) return;
    const int k = DIR > 0 ? j : pd.nt - 1 - j;
    const int ch = blockIdx.x;
    const real *S = A + pd.aoff;
    const int64_t ld = pd.ld;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    real *out = part + ((int64_t)blockIdx.y * nchunk + ch) * TILE;
    __shared__ real red[2 * TILE];
    if (DIR > 0) {
        const int64_t span = (int64_t)k * TILE;
        const int64_t per = ((span + nchunk - 1) / nchunk + 15) & ~(int64_t)15;
        const int64_t cb = ch * per, ce = min(span, cb + per);
        const int row = tid & 127, hf = tid >> 7;
        const real *Lr = S + (int64_t)k * TILE + row;
        const real *zz = vec + pd.yoff;
        real s0 = 0, s1 = 0;
        for (int64_t c = cb + 8 * hf; c < ce; c += 16) {       // each half takes alternate groups of 8 columns (chunks are
            real lv[8];                                         // multiples of 16 columns)
#pragma unroll
            for (int u = 0; u < 8; ++u) lv[u] = Lr[(c + u) * ld];
#pragma unroll
            for (int u = 0; u < 8; u += 2) { s0 += lv[u] * zz[c + u]; s1 += lv[u + 1] * zz[c + u + 1]; }
        }
        red[tid] = s0 + s1;
        __syncthreads();
        if (tid < TILE) out[tid] = red[tid] + red[tid + TILE];
    } else {
        const int64_t i0 = (int64_t)(k + 1) * TILE, span = ld - i0;
        const int64_t per = ((span + nchunk - 1) / nchunk + 63) & ~(int64_t)63;
        const int64_t ib = i0 + ch * per, ie = min((int64_t)ld, ib + per);
        const real *cs = vec + pd.yoff;
        for (int cc = wave; cc < TILE; cc += 4) {               // one wave per column, coalesced rows
            const real *col = S + ((int64_t)k * TILE + cc) * ld;
            real s0 = 0, s1 = 0;
            int64_t i = ib + lane;
            for (; i + 64 < ie; i += 128) { s0 += col[i] * cs[i]; s1 += col[i + 64] * cs[i + 64]; }
            for (; i < ie; i += 64) s0 += col[i] * cs[i];
            real s = s0 + s1;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0) out[cc] = s;
        }
    }
}

// z_s = -Ninv_s (v_s - sum_{jb<s} L[s][jb] z_jb), s = 0..3: the forward substitution of one diagonal tile (operands as in
// diag_solve_back)
__device__ __forceinline__ void diag_solve_fwd(const real *tri, real *v, real *w, int tid)
{
    const int sb = tid >> 5, i = tid & 31;
    for (int s2 = 0; s2 < 4; ++s2) {
        if (tid < TILE && sb == s2) {
            real t = v[32 * s2 + i];
            for (int jb = 0; jb < s2; ++jb) {
                const real *blk = tri + 1024 * (s2 * (s2 - 1) / 2 + jb);
                for (int c = 0; c < SB; ++c) t -= blk[i + 32 * c] * w[32 * jb + c];
            }
            v[32 * s2 + i] = t;
        }
        __syncthreads();
        if (tid < TILE && sb == s2) {
            const real *nb = tri + 1024 * (6 + s2);
            real t = 0;
            for (int c = 0; c <= i; ++c) t -= nb[i + 32 * c] * v[32 * s2 + c];      // -Ninv = L_ss^-1 (lower)
            w[32 * s2 + i] = t;
        }
        __syncthreads();
    }
}

// c_s = -Ninv_s^T (v_s - sum_{jb>s} L[jb][s]^T c_jb), s = 3..0: the backward substitution of one 128 x 128 diagonal tile
// whose operands stage_tri_operands left in `tri`; right-hand side in v, solution in w (LDS).  Entered by every thread of
// the workgroup (threads 0..127 work).
__device__ __forceinline__ void diag_solve_back(const real *tri, real *v, real *w, int tid)
{
    const int sb = tid >> 5, i = tid & 31;
    for (int s2 = 3; s2 >= 0; --s2) {
        if (tid < TILE && sb == s2) {
            real t = v[32 * s2 + i];
            for (int jb = s2 + 1; jb < 4; ++jb) {
                const real *blk = tri + 1024 * (jb * (jb - 1) / 2 + s2) + 32 * i;     // column i of block (jb, s2)
                for (int r = 0; r < SB; ++r) t -= blk[r] * w[32 * jb + r];
            }
            v[32 * s2 + i] = t;
        }
        __syncthreads();
        if (tid < TILE && sb == s2) {
            const real *nb = tri + 1024 * (6 + s2) + 32 * i;                           // column i of Ninv_s
            real t = 0;
            for (int r = i; r < SB; ++r) t -= nb[r] * v[32 * s2 + r];
            w[32 * s2 + i] = t;
        }
        __syncthreads();
    }
}

template <int DIR>
__global__ __launch_bounds__(256) void solve_block_kernel(const PatchDesc *__restrict__ descs, int j, int nchunk,
                                                          const real *__restrict__ A, const real *__restrict__ ninv,
                                                          const real *__restrict__ rhs, const real *__restrict__ part,
                                                          real *__restrict__ sol)
{
    const PatchDesc pd = descs[blockIdx.x];
    if (j >= pd.nt) return;
    const int k = DIR > 0 ? j : pd.nt - 1 - j;
    __shared__ real tri[TRI_LDS_DOUBLES];
    __shared__ real v[TILE], w[TILE];
    const int tid = threadIdx.x;
    const real *S = A + pd.aoff;
    const int64_t ld = pd.ld, d0 = (int64_t)k * TILE;
    stage_tri_operands(tri, S + d0 + d0 * ld, ld, ninv + pd.ioff + (int64_t)k * (4 * SB * SB), tid, 256);
    if (tid < TILE) {
        real acc = rhs[pd.yoff + d0 + tid];
        const bool any = DIR > 0 ? k > 0 : k + 1 < pd.nt;
        if (any)
            for (int c = 0; c < nchunk; ++c) acc -= part[((int64_t)blockIdx.x * nchunk + c) * TILE + tid];
        v[tid] = acc;
    }
    __syncthreads();
    const int sb = tid >> 5, i = tid & 31;          // threads 0..127: block sb, entry i
    if (DIR > 0) {
        diag_solve_fwd(tri, v, w, tid);
    } else {
        diag_solve_back(tri, v, w, tid);
    }
    if (tid < TILE) sol[pd.yoff + d0 + tid] = w[tid];
}

// ---------------------------------------------------------------------------------------------
// The triangular solves of the split path as ONE launch each (few, long patches: the block-by-block launches above cost
// two launch latencies per block).  DIR = +1: z = L^-1 y, workgroup b of a patch owns block k = b and needs the blocks
// above it; DIR = -1: c = L^-T z, workgroup b owns block k = nt - 1 - b and needs the blocks below it.  Either way a
// workgroup waits only for workgroups with LOWER ids, which were dispatched earlier: progress by construction, whatever
// fits on the chip.  It folds the tiles of its block row (forward) / block column (backward) into its right-hand side
// in a fixed order as the solution blocks appear -- one flag word per block holding the launch's epoch (no clearing
// between fits) -- then solves its diagonal tile and publishes its block: write-through stores, drained, barrier, flag
// (put1).  A wave owns 16 columns of every tile and reads them in 1 KiB row segments; every wave polls for itself, the
// next tile is in flight while it waits, partial sums stay in the lanes until the end.  Solution blocks are read with
// sc1 loads behind the matched poll (nobody reads a block before its flag: no stale line can exist).  Spins are
// bounded; a time-out raises the error word.  The summation order of a block is a function of the block alone.
// ---------------------------------------------------------------------------------------------
constexpr int CHAIN_THREADS = 512;
template <int DIR>
__global__ __launch_bounds__(CHAIN_THREADS) void solve_chain_kernel(const PatchDesc *__restrict__ descs,
                                                                    const real *__restrict__ A, const real *__restrict__ ninv,
                                                                    const real *__restrict__ rhs, real *sol, int32_t *flags,
                                                                    int32_t *err, int epoch, int max_nt)
{
    const PatchDesc pd = descs[blockIdx.y];
    if ((int)blockIdx.x >= pd.nt) return;
    const int k = DIR > 0 ? (int)blockIdx.x : pd.nt - 1 - (int)blockIdx.x;
    int32_t *fl = flags + (int64_t)blockIdx.y * max_nt;
    __shared__ real tri[TRI_LDS_DOUBLES];
    __shared__ real v[TILE], w[TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const real *S = A + pd.aoff;
    const int64_t ld = pd.ld, d0 = (int64_t)k * TILE;
    constexpr int NW = CHAIN_THREADS / 64, NC = TILE / NW;            // waves, columns of a tile per wave
    stage_tri_operands(tri, S + d0 + d0 * ld, ld, ninv + pd.ioff + (int64_t)k * (4 * SB * SB), tid, CHAIN_THREADS);
    // tile t of this block's sweep: forward (k, t), columns 128 t + ..., rows d0 + ...; backward (t, k), columns d0 + ...,
    // rows 128 t + ...; a lane holds rows 2 lane, 2 lane + 1 of the wave's NC columns
    const real *base = DIR > 0 ? S + (int64_t)(NC * wave) * ld + d0 + 2 * lane : S + (d0 + NC * wave) * ld + 2 * lane;
    const int64_t tstep = DIR > 0 ? (int64_t)TILE * ld : (int64_t)TILE;
    real *ss = sol + pd.yoff;
    const int t_first = DIR > 0 ? 0 : pd.nt - 1, t_end = k, dt = DIR > 0 ? 1 : -1;     // t = t_first, t_first + dt, ... != k
    real2_t sum[NC], nxt[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) sum[j] = real2_t{0, 0};
    if (t_first != t_end) {
#pragma unroll
        for (int j = 0; j < NC; ++j)
            nxt[j] = __builtin_nontemporal_load(reinterpret_cast<const real2_t *>(base + j * ld + t_first * tstep));
    }
    for (int t = t_first; t != t_end; t += dt) {
        real2_t cur[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) cur[j] = nxt[j];
        if (t + dt != t_end) {
#pragma unroll
            for (int j = 0; j < NC; ++j)
                nxt[j] = __builtin_nontemporal_load(reinterpret_cast<const real2_t *>(base + j * ld + (t + dt) * tstep));
        }
        // block t of the solution is there?  (every wave for itself: no barrier in this loop)
        if (__hip_atomic_load(fl + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                __builtin_amdgcn_s_sleep(4);
                if (__hip_atomic_load(fl + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) break;
                if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                if (__builtin_amdgcn_s_memrealtime() - t0 > QUEUE_SPIN_TICKS) {
                    if (lane == 0) __hip_atomic_store(err, 1 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        asm volatile("" ::: "memory");
        if (DIR > 0) {
            // the wave's NC entries of z_t, one per lane (lanes 0..NC-1), broadcast column by column
            const real zl = __hip_atomic_load(ss + (int64_t)t * TILE + NC * wave + (lane & (NC - 1)), __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const real zj = __shfl(zl, j);
                sum[j & 3] += cur[j] * real2_t{zj, zj};
            }
        } else {
            real2_t ci;
            ci[0] = __hip_atomic_load(ss + (int64_t)t * TILE + 2 * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ci[1] = __hip_atomic_load(ss + (int64_t)t * TILE + 2 * lane + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int j = 0; j < NC; ++j) sum[j] += cur[j] * ci;
        }
    }
    if (DIR > 0) {
        // rows live in the lanes: the waves' partial sums meet in LDS (tri is not touched before the barrier below... it is
        // being staged: use a separate array)
        __shared__ real red[NW * TILE];
        const real2_t r = (sum[0] + sum[1]) + (sum[2] + sum[3]);
        *reinterpret_cast<real2_t *>(red + wave * TILE + 2 * lane) = r;
        __syncthreads();
        if (tid < TILE) {
            real acc = rhs[pd.yoff + d0 + tid];
#pragma unroll
            for (int u = 0; u < NW; ++u) acc -= red[u * TILE + tid];
            v[tid] = acc;
        }
    } else {
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            real r = sum[j][0] + sum[j][1];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) r += __shfl_xor(r, o);
            if (lane == 0) v[NC * wave + j] = rhs[pd.yoff + d0 + NC * wave + j] - r;
        }
    }
    __syncthreads();
    if (DIR > 0) diag_solve_fwd(tri, v, w, tid);
    else diag_solve_back(tri, v, w, tid);
    if (tid < TILE) put1<true>(ss + d0 + tid, w[tid]);
    drain_stores();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(fl + k, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
        // r[col] = z_k[col] - sum_{i >= d0 + TILE} L[i, d0 + col] c[i]: a wave takes two columns at a time (cc, cc + 16)
        // and four 128-row chunks of each, 16-byte loads: eight 1 KiB loads in flight per wave (the loop used to keep
        // four 512-byte loads in flight and wait for all of them every pass -- latency bound at 4 TB/s)
        const int64_t i0 = d0 + TILE;
        const int nrow = (int)(ld - i0);                       // a multiple of TILE
        for (int cc = wave; cc < TILE; cc += 32) {
            const real *colA = S + (d0 + cc) * ld + i0, *colB = colA + 16 * ld;
            const real *cp = cs + i0;
            real2_t sA[4], sB[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) sA[u] = sB[u] = real2_t{0, 0};
            int i = 2 * lane;
            for (; i + 384 < nrow; i += 512) {
                real2_t a[4], b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a[u] = __builtin_nontemporal_load(reinterpret_cast<const real2_t *>(colA + i + 128 * u));
                    b[u] = __builtin_nontemporal_load(reinterpret_cast<const real2_t *>(colB + i + 128 * u));
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const real2_t c = *reinterpret_cast<const real2_t *>(cp + i + 128 * u);
                    sA[u] += a[u] * c;
                    sB[u] += b[u] * c;
                }
            }
            for (; i < nrow; i += 128) {
                const real2_t a = __builtin_nontemporal_load(reinterpret_cast<const real2_t *>(colA + i));
                const real2_t b = __builtin_nontemporal_load(reinterpret_cast<const real2_t *>(colB + i));
                const real2_t c = *reinterpret_cast<const real2_t *>(cp + i);
                sA[0] += a * c;
                sB[0] += b * c;
            }
            const real2_t tA = (sA[0] + sA[1]) + (sA[2] + sA[3]), tB = (sB[0] + sB[1]) + (sB[2] + sB[3]);
            real vA = tA[0] + tA[1], vB = tB[0] + tB[1];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                vA += __shfl_xor(vA, o);
                vB += __shfl_xor(vB, o);
            }
            if (lane == 0) {
                r[cc] = z[pd.yoff + d0 + cc] - vA;
                r[cc + 16] = z[pd.yoff + d0 + cc + 16] - vB;
            }
        }
        __syncthreads();
        // c_k = L[kk]^-T r by block backward substitution with the negated inverted 32 x 32 blocks, all from LDS.
        // Both products of a stage are spread over the 16 waves: a wave owns columns 2 wave and 2 wave + 1 (one per
        // half wave), a lane one row of the 32-row block, and the 32 partial products meet in a shuffle reduction
        // (one thread per column walking 96 + 32 LDS entries was ~10 us of serial work per block row).
        const int hcol = 2 * wave + (lane >> 5), hl = lane & 31;
        for (int s = 3; s >= 0; --s) {
            {
                real v = 0.0;
                for (int bi = s + 1; bi < 4; ++bi)
                    v += Lt[(bi * (bi - 1) / 2 + s) * (SB * LDN) + hcol * LDN + hl] * cs[d0 + SB * bi + hl];
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
                if (hl == 0 && s < 3) r[SB * s + hcol] -= v;
            }
            __syncthreads();
            {
                // (D^-T r)[hcol] = sum_{i >= hcol} Ninv[i][hcol] r[i], negated inverse stored
                real v = hl >= hcol ? Nt[s * (SB * LDN) + hl + hcol * LDN] * r[SB * s + hl] : (real)0;
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
                if (hl == 0) cs[d0 + SB * s + hcol] = -v;
            }
            __threadfence_block();
            __syncthreads();
        }
    }
    if (cs_in_lds)
        for (int i = tid; i < pd.ld; i += 1024) cvec[pd.yoff + i] = cs[i];
}

// workspace of the split path, grow only
static int reserve_split(pmk_model *m, size_t partial_bytes, size_t solve_bytes)
{
    if (partial_bytes > m->partial_bytes) {
        if (m->d_partial) PMK_HIP(hipFree(m->d_partial));
        m->d_partial = nullptr; m->partial_bytes = 0;
        PMK_HIP(hipMalloc(&m->d_partial, partial_bytes));
        m->partial_bytes = partial_bytes;
    }
    if (solve_bytes > m->solve_bytes) {
        if (m->d_solve_part) PMK_HIP(hipFree(m->d_solve_part));
        m->d_solve_part = nullptr; m->solve_bytes = 0;
        PMK_HIP(hipMalloc(&m->d_solve_part, solve_bytes));
        m->solve_bytes = solve_bytes;
    }
    return 0;
}

// task lists of the queue path (see chol_queue_kernel), built once per model: list x serves the patch slots x, x + 8, ...
// of `order` (sizes interleaved over the XCDs, all block rows of a patch on one XCD) in the order of the end-aligned step
// schedule: for every step l the critical tasks first, then the block rows that are critical at step l + 1, then the
// remaining block rows patch by patch (they stream the same block row k of L at the same time)
static int build_queue(pmk_model *m)
{
    const int P = (int)m->P, max_nt = m->max_nt;
    std::vector<CholTask> all;
    int nseg = 1;
    if (const char *e = std::getenv("PMK_QUEUE_SEGS")) nseg = std::max(1, std::min(QUEUE_MAX_SEGS, std::atoi(e)));
    // the queue takes over at step L0 (the steps before it run as one launch each, launch_cholesky); negative: counted
    // from the end
    int L0 = m->queue_from;
    if (L0 < 0) L0 = std::max(0, max_nt - 1 + L0);
    L0 = std::min(L0, std::max(0, max_nt - 1));
    m->queue_l0 = L0;
    const int nsteps = max_nt - 1 - L0;
    nseg = std::min(nseg, std::max(1, nsteps));
    m->qsegs = nseg;
    m->qoff.assign((size_t)nseg * 9, 0);
    for (int sg = 0; sg < nseg; ++sg)
        for (int x = 0; x < 8; ++x) {
            const int l_lo = L0 + (int)((int64_t)nsteps * sg / nseg), l_hi = L0 + (int)((int64_t)nsteps * (sg + 1) / nseg);
            auto col = [&](int pid, int l) { return l - (max_nt - m->desc[(size_t)pid].nt); };
            auto nact = [&](int l) { return m->active_prefix[(size_t)std::min(max_nt + 1, max_nt - l)]; };
            // ONE list per XCD is in use (list 2 x; list 2 x + 1 stays empty), in an order in which everything a task waits
            // for comes earlier in the same list -- step by step: the block row the chain waits for (row k + 1), the
            // look-ahead of the next diagonal tile, its potrf, the block row after that, then the rest patch by patch (they
            // stream the same block row k of L at the same time).  A task is then only ever taken after everything it
            // depends on has been taken, by a workgroup that is running: progress by construction.  (A chain list and a
            // rows list per XCD, served by the two workgroups of a CU with opposite preference, pairs every potrf with a
            // GEMM on its CU and was measured no faster; its take-before-ready protocol is not in the tree.)
            std::vector<CholTask> cl;
            if (sg == 0 && L0 == 0)
                for (int s = x; s < P; s += 8) cl.push_back({m->order[(size_t)s], 0, 0, QT_POT, 0});
            for (int l = l_lo; l < l_hi; ++l) {
                const int G = max_nt - l - 1;
                for (int s = x; s < nact(l); s += 8) {
                    const int pid = m->order[(size_t)s], k = col(pid, l);
                    cl.push_back({pid, (int16_t)k, (int16_t)(k + 1), QT_ROW, 0});
                }
                for (int s = x; s < nact(l); s += 8) {
                    const int pid = m->order[(size_t)s], k = col(pid, l);
                    if (k >= 1) cl.push_back({pid, (int16_t)(k + 1), (int16_t)(k + 1), QT_LOOK, 0});
                }
                for (int s = x; s < nact(l); s += 8) {
                    const int pid = m->order[(size_t)s], k = col(pid, l);
                    cl.push_back({pid, (int16_t)(k + 1), (int16_t)(k + 1), QT_POT, 0});
                }
                if (G >= 2)
                    for (int s = x; s < nact(l); s += 8) {
                        const int pid = m->order[(size_t)s], k = col(pid, l);
                        cl.push_back({pid, (int16_t)k, (int16_t)(k + 2), QT_ROW, 0});
                    }
                for (int s = x; s < nact(l); s += 8) {
                    const int pid = m->order[(size_t)s], k = col(pid, l);
                    for (int bx = 2; bx < G; ++bx) cl.push_back({pid, (int16_t)k, (int16_t)(k + 1 + bx), QT_ROW, 0});
                }
            }
            m->qoff[(size_t)sg * 9 + x] = (int32_t)all.size();
            all.insert(all.end(), cl.begin(), cl.end());
            m->qoff[(size_t)sg * 9 + x + 1] = (int32_t)all.size();
        }
    m->qfstride = (max_nt + 2 + 15) & ~15;       // diagonal count, one word per block row, the look-ahead word (last)
    m->sched_bytes = sizeof(int32_t) * ((size_t)SCHED_FLAGS + (size_t)P * (size_t)m->qfstride);
    // state of the flags when the queue takes over: a patch that is at column k0 of its own schedule at step L0 has its
    // tiles 0 .. k0 factorised and every block row final through column k0 - 1 (k0 = 0: the first tile, chol_first_kernel)
    std::vector<int32_t> init((size_t)SCHED_FLAGS + (size_t)P * (size_t)m->qfstride, 0);
    if (L0 > 0)
        for (int pid = 0; pid < P; ++pid) {
            const int k0 = std::max(0, L0 - (max_nt - m->desc[(size_t)pid].nt));
            int32_t *fl = init.data() + SCHED_FLAGS + (size_t)pid * (size_t)m->qfstride;
            fl[0] = k0 + 1;
            for (int i = 0; i < m->desc[(size_t)pid].nt; ++i) fl[1 + i] = k0;
            fl[m->qfstride - 1] = k0;           // the look-ahead word: tiles <= k0 need none any more
        }
    PMK_HIP(hipMalloc(&m->d_qtasks, sizeof(CholTask) * std::max<size_t>(all.size(), 1)));
    PMK_HIP(hipMalloc((void **)&m->d_sched, m->sched_bytes));
    PMK_HIP(hipMalloc((void **)&m->d_sched_init, m->sched_bytes));
    PMK_HIP(hipMemcpy(m->d_sched_init, init.data(), m->sched_bytes, hipMemcpyHostToDevice));
    PMK_HIP(hipMemcpy(m->d_qtasks, all.data(), sizeof(CholTask) * all.size(), hipMemcpyHostToDevice));
    m->queue_built = true;
    return 0;
}

static int launch_cholesky_queue(pmk_model *m, hipStream_t s)
{
    pmk_ctx *c = m->ctx;
    PMK_HIP(hipMemcpyAsync(m->d_sched, m->d_sched_init, m->sched_bytes, hipMemcpyDeviceToDevice, s));
    const bool fine = c->timers >= 2;
    for (int sg = 0; sg < m->qsegs; ++sg) {
        const int evi = m->queue_l0 + sg;
        if (fine) {
            while ((int)c->panel_ev.size() <= evi) {
                hipEvent_t a, b;
                PMK_HIP(hipEventCreate(&a));
                PMK_HIP(hipEventCreate(&b));
                c->panel_ev.push_back({a, b});
            }
            PMK_HIP(hipEventRecord(c->panel_ev[(size_t)evi].first, s));
        }
        QueueOffsets qo;
        for (int x = 0; x < 9; ++x) qo.off[x] = m->qoff[(size_t)sg * 9 + x];
        const unsigned grid = (unsigned)std::min<int64_t>(2 * (int64_t)c->num_cu, std::max<int64_t>(8, qo.off[8] - qo.off[0]));
        hipLaunchKernelGGL(chol_queue_kernel, dim3(grid), dim3(256), 0, s, m->d_desc, (const CholTask *)m->d_qtasks, qo,
                           m->d_sched, sg, m->qfstride, (real *)m->d_a, (real *)m->d_inv, (const real *)m->d_y,
                           (real *)m->d_z, m->d_info, sg == 0 && m->queue_l0 == 0 ? m->ctx->d_clk : nullptr);
        if (fine) {
            PMK_HIP(hipEventRecord(c->panel_ev[(size_t)evi].second, s));
            c->panel_n = evi + 1;
        }
    }
    PMK_HIP(hipGetLastError());
    m->queue_used = true;
    return 0;
}

// col_ev (may be null): event s of the pipelined kernel-matrix build = "stage s is in the slabs"; launch l reads block
// columns that stages <= l + 1 produced (pmk_kmat.hip), the first kernel reads stage 0
int launch_cholesky(pmk_model *m, hipStream_t s, int64_t p0, int64_t np, const hipEvent_t *col_ev, int n_ev)
{
    // gemm_nt consumes K in groups of 4*PF k-indices; K is always a multiple of TILE here
    static_assert(TILE % (4 * PF_DIAG) == 0 && TILE % (4 * PF_CHOL) == 0 && TILE % (4 * PFJ_CHOL) == 0,
                  "prefetch depth must divide TILE/4");
    if (p0 != 0 || np != m->P) { set_error("launch_cholesky: sub-batches are not supported"); return -2; }
    pmk_ctx *c = m->ctx;
    c->panel_n = 0;
    m->queue_used = false;
    // the task queue takes the steps from m->queue_l0 on (0: all of them, and the first tiles too)
    int l_end = m->max_nt - 1;
    if (m->queue_mode && !m->split_mode && !col_ev) {
        if (!m->queue_built)
            if (int rc = build_queue(m)) return rc;
        l_end = m->queue_l0;
        if (l_end == 0) {
            // no chol_first_kernel in front: clear the status words and the clock stamps here
            PMK_HIP(hipMemsetAsync(m->d_info, 0, sizeof(int32_t) * (size_t)np, s));
            for (int x = 0; x < 8; ++x)
                PMK_HIP(hipMemsetAsync(m->ctx->d_clk + 130 * x, 0, sizeof(unsigned long long) * 128, s));
            return launch_cholesky_queue(m, s);
        }
    }
    const bool fine = c->timers >= 2;
    const int want_wg = 2 * c->num_cu;                     // workgroups that fill the chip (two per CU)
    auto ev_begin = [&](int l) -> int {
        if (!fine) return 0;
        while ((int)c->panel_ev.size() <= l) {
            hipEvent_t a, b;
            PMK_HIP(hipEventCreate(&a));
            PMK_HIP(hipEventCreate(&b));
            c->panel_ev.push_back({a, b});
        }
        PMK_HIP(hipEventRecord(c->panel_ev[(size_t)l].first, s));
        return 0;
    };
    auto ev_end = [&](int l) -> int {
        if (!fine) return 0;
        PMK_HIP(hipEventRecord(c->panel_ev[(size_t)l].second, s));
        c->panel_n = l + 1;
        return 0;
    };
    const int l0 = 0;
    // split path: K chunks so that ONE patch's tiles x chunks would fill the chip (a function of the step alone, not of the
    // number of patches: the summation order of a tile, hence every bit of the factor, is then the same whether a patch
    // is factorised alone or next to others -- sharded and single models stay bit-identical); a chunk is at least two
    // block columns deep; 64-way: the serial sum in the combine costs more than it buys
    auto nsplit_of = [&](int l, int G) {
        return (m->split_mode && l >= 4) ? std::max(1, std::min(std::min(16, l / 2), (want_wg + G) / (G + 1))) : 1;
    };
    // the partial tiles of a step live in one half of the buffer, those of the next step in the other: the diagonal tile a
    // split step leaves behind is factorised by the first workgroups of the NEXT partial launch, from the previous half
    size_t half_tiles = 0;
    for (int l = l0; l < l_end; ++l) {
        const int na = m->active_prefix[(size_t)std::min(m->max_nt + 1, m->max_nt - l + l0)], G = m->max_nt - l - 1;
        const int ns = nsplit_of(l, G);
        if (ns > 1) half_tiles = std::max(half_tiles, (size_t)na * (size_t)(G + 1) * (size_t)ns);
    }
    if (half_tiles)
        if (int rc = reserve_split(m, 2 * sizeof(real2_t) * (size_t)PARTIAL_TILE * half_tiles, 0)) return rc;
    static const bool prereduce = !std::getenv("PMK_SPLIT_PREREDUCE") || std::atoi(std::getenv("PMK_SPLIT_PREREDUCE")) != 0;
    struct Pending { int n, G, nsplit; const real2_t *buf; int l; } pend = {0, 0, 1, nullptr, -1};
    auto flush_pending = [&]() -> int {       // a potrf-only launch (no split step follows the one that left it)
        if (pend.n == 0) return 0;
        hipLaunchKernelGGL(chol_partial_kernel, dim3((unsigned)pend.n), dim3(256), 0, s, m->d_desc, m->d_order, 0, 0, 1,
                           pend.l + 1, m->max_nt, (real *)m->d_a, (real2_t *)nullptr, pend.n, pend.G, pend.nsplit, pend.buf,
                           (real *)m->d_inv, (const real *)m->d_y, (real *)m->d_z, m->d_info);
        pend.n = 0;
        return 0;
    };
    if (col_ev && n_ev > 0) PMK_HIP(hipStreamWaitEvent(s, col_ev[0], 0));
    hipLaunchKernelGGL(chol_first_kernel, dim3((unsigned)np), dim3(256), 0, s, m->d_desc, (real *)m->d_a, (real *)m->d_inv,
                       (const real *)m->d_y, (real *)m->d_z, m->d_info, m->ctx->d_clk);
    for (int l = l0; l < l_end; ++l) {
        // patch p runs block column k = l - (max_nt - nt_p) at launch l (end-aligned); it takes part from k = l0 on:
        // nt_p >= max_nt - l + l0.  Those patches are a prefix of `order` (sorted by nt, largest first).
        const int nactive = m->active_prefix[(size_t)std::min(m->max_nt + 1, m->max_nt - l + l0)];
        const int G = m->max_nt - l - 1;                   // block rows below the diagonal, the same for every active patch
        if (col_ev && l + 1 < n_ev) PMK_HIP(hipStreamWaitEvent(s, col_ev[l + 1], 0));
        if (nactive == 0) continue;
        if (int rc = ev_begin(l)) return rc;
        const int nsplit = nsplit_of(l, G);
        const unsigned grid = (unsigned)(8 * ((nactive + 7) / 8) * G);
        if (nsplit > 1) {
            const int tiles = G + 1;
            real2_t *buf = (real2_t *)m->d_partial + (size_t)(l & 1) * half_tiles * PARTIAL_TILE;     // alternate halves
            hipLaunchKernelGGL(chol_partial_kernel, dim3((unsigned)(pend.n + nactive * tiles * nsplit)), dim3(256), 0, s,
                               m->d_desc, m->d_order, nactive, G, nsplit, l, m->max_nt, (real *)m->d_a, buf, pend.n, pend.G,
                               pend.nsplit, pend.buf, (real *)m->d_inv, (const real *)m->d_y, (real *)m->d_z, m->d_info);
            // Fold the next diagonal tile's partial sums into the slab beside this step's block rows (one more workgroup per
            // patch)?  Only where the workgroup that factorises it is what the NEXT partial launch waits for: its products
            // are short (<= 4 block columns per chunk).  Measured: n = 8192 15.5 -> 14.0 ms; at 16384 and beyond the
            // products are the longer leg and the extra workgroup only stretches this launch (40.0 -> 43.7 ms if always on).
            const bool fold = prereduce && G >= 2 && (l + 1) <= 4 * nsplit_of(l + 1, G - 1);
            hipLaunchKernelGGL((chol_step_kernel<1, 0>), dim3(grid + (fold ? (unsigned)nactive : 0u)), dim3(256), 0, s, m->d_desc, m->d_order, nactive, G, l,
                               m->max_nt, (real *)m->d_a, (real *)m->d_inv, (const real *)m->d_y, (real *)m->d_z, m->d_info,
                               (const real2_t *)buf, nsplit, m->ctx->d_clk, (const real *)m->d_x, m->th, fold ? nsplit : 0);
            pend = {nactive, G, fold ? 0 : nsplit, buf, l};                // tile k + 1 of these patches: with the next launch (its partial
                                                           // tiles are folded in by the step launch above: none left)
        } else {
            if (int rc = flush_pending()) return rc;
#define PMK_STEP(KD_)                                                                                                        \
            hipLaunchKernelGGL((chol_step_kernel<0, KD_>), dim3(grid), dim3(256), 0, s, m->d_desc, m->d_order, nactive, G, l,  \
                               m->max_nt, (real *)m->d_a, (real *)m->d_inv, (const real *)m->d_y, (real *)m->d_z, m->d_info,  \
                               (const real2_t *)nullptr, 1, m->ctx->d_clk, (const real *)m->d_x, m->th, 0)
            // fused kernel-matrix build (pmk_model_fit decides): the strictly lower tiles are evaluated at their first use
            if (m->fuse_k1 && m->D == 2) PMK_STEP(2);
            else if (m->fuse_k1 && m->D == 3) PMK_STEP(3);
            else PMK_STEP(0);
#undef PMK_STEP
        }
        if (int rc = ev_end(l)) return rc;
    }
    if (int rc = flush_pending()) return rc;
    PMK_HIP(hipGetLastError());
    if (l_end < m->max_nt - 1) return launch_cholesky_queue(m, s);
    return 0;
}

// the two triangular solves of the split path: z = L^-1 y, then c = L^-T z, block by block with the long products cut
// into chunks (see solve_partial_kernel)
static int launch_split_solves(pmk_model *m, hipStream_t s)
{
    const int P = (int)m->P;
    // one chained launch per solve when the patches' blocks (nearly) fit on the chip together, else block by block
    static const char *chain_env = std::getenv("PMK_SPLIT_CHAIN");
    const bool chain = m->chain_mode >= 0 ? m->chain_mode != 0
                       : chain_env ? std::atoi(chain_env) != 0 : (int64_t)P * m->max_nt <= 2 * (int64_t)m->ctx->num_cu;
    m->chain_used = chain;
    if (chain) {
        const size_t words = 16 + 2 * (size_t)P * (size_t)m->max_nt;          // error word, flags of z, flags of c
        if (words > m->chain_words) {
            if (m->d_chain) PMK_HIP(hipFree(m->d_chain));
            m->d_chain = nullptr; m->chain_words = 0;
            PMK_HIP(hipMalloc(&m->d_chain, sizeof(int32_t) * words));
            m->chain_words = words;
            m->chain_epoch = INT32_MAX;
        }
        if (m->chain_epoch == INT32_MAX) {
            PMK_HIP(hipMemsetAsync(m->d_chain, 0, sizeof(int32_t) * m->chain_words, s));
            m->chain_epoch = 0;
        }
        ++m->chain_epoch;
        int32_t *w = (int32_t *)m->d_chain;
        const dim3 grid((unsigned)m->max_nt, (unsigned)P);
        hipLaunchKernelGGL(solve_chain_kernel<1>, grid, dim3(CHAIN_THREADS), 0, s, m->d_desc, (const real *)m->d_a,
                           (const real *)m->d_inv, (const real *)m->d_y, (real *)m->d_z, w + 16, w, m->chain_epoch, m->max_nt);
        hipLaunchKernelGGL(solve_chain_kernel<-1>, grid, dim3(CHAIN_THREADS), 0, s, m->d_desc, (const real *)m->d_a,
                           (const real *)m->d_inv, (const real *)m->d_z, (real *)m->d_c, w + 16 + (size_t)P * (size_t)m->max_nt, w,
                           m->chain_epoch, m->max_nt);
        PMK_HIP(hipGetLastError());
        return 0;
    }
    const int max_chunks = 128;      // a function of the block alone, not of P: bit-identical results however patches are grouped
    if (int rc = reserve_split(m, 0, sizeof(real) * (size_t)P * max_chunks * TILE)) return rc;
    real *part = (real *)m->d_solve_part;
    for (int dir = 0; dir < 2; ++dir) {
        const real *rhs = dir == 0 ? (const real *)m->d_y : (const real *)m->d_z;
        real *sol = dir == 0 ? (real *)m->d_z : (real *)m->d_c;
        for (int j = 0; j < m->max_nt; ++j) {
            const int nchunk = std::max(1, std::min(max_chunks, j * TILE / 256));
            if (j > 0) {
                if (dir == 0)
                    hipLaunchKernelGGL(solve_partial_kernel<1>, dim3((unsigned)nchunk, (unsigned)P), dim3(256), 0, s, m->d_desc, j,
                                       nchunk, (const real *)m->d_a, (const real *)sol, part);
                else
                    hipLaunchKernelGGL(solve_partial_kernel<-1>, dim3((unsigned)nchunk, (unsigned)P), dim3(256), 0, s, m->d_desc, j,
                                       nchunk, (const real *)m->d_a, (const real *)sol, part);
            }
            if (dir == 0)
                hipLaunchKernelGGL(solve_block_kernel<1>, dim3((unsigned)P), dim3(256), 0, s, m->d_desc, j, nchunk,
                                   (const real *)m->d_a, (const real *)m->d_inv, rhs, part, sol);
            else
                hipLaunchKernelGGL(solve_block_kernel<-1>, dim3((unsigned)P), dim3(256), 0, s, m->d_desc, j, nchunk,
                                   (const real *)m->d_a, (const real *)m->d_inv, rhs, part, sol);
        }
    }
    PMK_HIP(hipGetLastError());
    return 0;
}

#if defined(PMK_QTRACE) && !defined(PMK_REAL_F32)
// diagnostic build only: the per-task stamps of the last queue launch and its task list (pid, k, row, type per task)
extern "C" int pmk_qtrace_dump(pmk_model *m, unsigned long long *stamps, int32_t *tasks, int ntasks_max, int32_t *qoff)
{
    const int nt = std::min(std::min(ntasks_max, (int)m->qoff.back()), QTRACE_MAX);
    for (int x = 0; x < 9; ++x) qoff[x] = m->qoff[(size_t)x];
    if (hipMemcpyFromSymbol(stamps, HIP_SYMBOL(g_qtrace), sizeof(unsigned long long) * QTRACE_WORDS * (size_t)nt) != hipSuccess) return -1;
    if (hipMemcpy(tasks, m->d_qtasks, sizeof(CholTask) * (size_t)nt, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return nt;
}
#endif

#if defined(PMK_TRACE) && !defined(PMK_REAL_F32)
extern "C" int pmk_trace_dump(unsigned long long *out, int nwords)
{
    const size_t want = sizeof(unsigned long long) * (size_t)std::min(nwords, TRACE_WORDS * TRACE_MAX_WG);
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), want) == hipSuccess ? 0 : -1;
}
#endif

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

// per-device kernel attributes (called by pmk_ctx_create with the context's device current): the back
// substitution always needs more than the default 64 KB of dynamic LDS
int set_device_attributes()
{
    PMK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(chol_backsolve_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return 0;
}

int launch_backsolve(pmk_model *m, hipStream_t s, int64_t p0, int64_t np)
{
    if (m->split_mode) return launch_split_solves(m, s);
    const size_t fixed = sizeof(real) * (TILE + 10 * SB * (SB + 1));
    const size_t csb = sizeof(real) * (size_t)m->max_nt * TILE;
    const int cs_in_lds = fixed + csb <= 150 * 1024;
    const size_t lds = fixed + (cs_in_lds ? csb : 0);
    hipLaunchKernelGGL(chol_backsolve_kernel, dim3((unsigned)np), dim3(1024), lds, s, m->d_desc + p0, (real *)m->d_a, (real *)m->d_inv,
                       (real *)m->d_z, (real *)m->d_c, cs_in_lds);
    PMK_HIP(hipGetLastError());
    return 0;
}

}  // namespace PMK_NS
}  // namespace pmk
