// K4: per-(query, region) prediction, queryinner! of the reference (src/RKHS/mixtureGP.jl:296-316)
// batched over all the queries that touch a region:
//     kq = k(xq, X_r)            (cross-kernel tile, evaluated in registers, never written to HBM)
//     u  = kq . c_r              (mean)
//     V  = L_r^-1 kq             (blocked forward substitution, every product on fp64 MFMA)
//     v  = clamp(k(xq,xq) - |V|^2, 1e-12, inf)
// The reference runs one dtrsv per (query, region) and streams the 4 n^2-byte factor each time; here a
// workgroup owns a strip of TQ = 256 query columns (8 waves x 32 columns: one workgroup per CU, two waves per
// SIMD) and sweeps the block rows of L once for all of them.
//
// Strip task, block row i:  acc(128 x 32) = Kq_i - sum_{j<i} L[i,j] V_j   (gemm_nt_indexed, V_j re-read from the
// wave's strip in global memory, which is stored negated so the MFMA accumulates the subtraction)
//                           -V_i = -L[ii]^-1 acc                          (tri_solve_global: block substitution,
//                                                                          operands prefetched from the factor)
// The schedule of a strip (what is kept off the critical path, and how) is described at the kernel.
//
// HBM traffic.  A strip streams the factor once (n^2/2 elements per TQ columns) and re-reads its own V block rows
// (n^2/(2*128) elements per column).  The factor is the operand every strip of a region shares: strips of one region
// that run at the same time on one XCD are kept in LOCK-STEP (a bounded spin on an arrival counter once per block
// row -- timing only, no data is handed over, so a missed rendezvous costs time, never correctness)
// and the block row of L is then served to all of them by ONE fetch into that XCD's L2.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <tuple>

#include "pmk_mfma.h"

namespace pmk {
namespace PMK_NS {

#ifndef PMK_PFI_PRED
#define PMK_PFI_PRED 4
#endif
constexpr int PF_PRED = PMK_PFI_PRED;       // I-operand (factor, from L2) prefetch depth in k-steps
#ifndef PMK_PFJ
#define PMK_PFJ 4
#endif
constexpr int PFJ_PRED = PMK_PFJ;   // J-operand (the wave's own strip columns, HBM) prefetch depth
// column pairs (of 32) per wave: 1 = eight waves of 128 x 32 tiles, two per SIMD (256 registers each); 2 = four waves of
// 128 x 64 tiles, ONE per SIMD with all 512 registers (accumulators in the AGPR half)
#ifndef PMK_PRED_NPJ
#define PMK_PRED_NPJ 1
#endif
constexpr int NPJW = PMK_PRED_NPJ;
constexpr int WCOLS = 32 * NPJW;            // query columns of a wave
constexpr int PRED_WAVES = TQ / WCOLS;
constexpr int PRED_THREADS = 64 * PRED_WAVES;
typedef real kvec_t __attribute__((ext_vector_type(2 * NPJW)));      // the kernel values of a lane's columns for one row
#ifndef PMK_SYNC_TICKS
#define PMK_SYNC_TICKS 4000         // lock-step rendezvous: give up after 40 us (s_memrealtime ticks of 10 ns)
#endif
#ifndef PMK_ROUND_TICKS
#define PMK_ROUND_TICKS 300000      // round barrier: give up after 3 ms, and then for the rest of the launch
#endif

#ifdef PMK_TRACE
// diagnostic build only: rendezvous statistics [0] syncs, [1] timeouts, [2] total wait ticks (10 ns), [3] max wait
__device__ unsigned long long g_sync_stats[4];
// phase stamps of block row PMK_TRACE_ROW in round PMK_TRACE_ROUND: [wg][wave][8]
__device__ unsigned long long g_row_stamp[512 * 40];
#ifndef PMK_TRACE_ROW
#define PMK_TRACE_ROW 8
#endif
#ifndef PMK_TRACE_ROUND
#define PMK_TRACE_ROUND 3
#endif
#define PMK_PSTAMP(k)                                                                                        \
    do {                                                                                                     \
        if (round == PMK_TRACE_ROUND && i == PMK_TRACE_ROW && lane == 0 && blockIdx.x < 64)                   \
            g_row_stamp[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (k)] = __builtin_amdgcn_s_memrealtime();               \
    } while (0)
#define PMK_TRACED (round == PMK_TRACE_ROUND)
#else
#define PMK_PSTAMP(k)
#define PMK_TRACED false
#endif

struct StripTask {
    int32_t region;    // local patch index in the model
    int32_t count;     // valid columns (<= TQ)
    int64_t first;     // first sorted item of the strip
    int32_t group;     // lock-step group (arrival counter index); -1: alone
    int32_t gsize;     // strips in the group
};

// one block row of the strip: acc = Kq_i (in) -> -V_i (out)
template <int NACT>
__device__ __forceinline__ void strip_block_row(WaveTile<4, NPJW> &acc, const real *Li, int64_t ld, const real *V, int i,
                                                const real *Lii, const real *ninv_i, int lane, bool traced = false)
{
    if (i > 0) {
        // order this wave's earlier strip stores before its loads of them
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#ifdef PMK_PRED_POINTER
        gemm_nt<4, NPJW, PF_PRED, PFJ_PRED, NACT>(acc, Li, ld, V, TQ, i * TILE, lane);
#elif defined(PMK_PRED_BUF)
        gemm_nt_buf<4, NPJW, PF_PRED, NACT>(acc, Li, ld, V, TQ, i * TILE, lane);
#else
        gemm_nt_indexed<4, NPJW, PF_PRED, PFJ_PRED, NACT>(acc, Li, ld, V, TQ, i * TILE, lane);
#endif
    }
#ifdef PMK_TRACE
    if (traced && i == PMK_TRACE_ROW && lane == 0 && blockIdx.x < 64) {
        g_row_stamp[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + 3] = __builtin_amdgcn_s_memrealtime();
        g_row_stamp[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + 7] = __builtin_amdgcn_s_memtime();      // shader clock
    }
#endif
    // the TRSM operands come straight from the factor (prefetched block by block into registers): no LDS copy, no
    // barrier around staging one
    tri_solve_global<NPJW>(acc, Lii, ld, ninv_i, lane);
}

// Kq tile of one block row (query is the first kernel argument, mixtureGP.jl:304) into the wave's accumulator, 32 rows
// at a time in a ROLLED loop whose results pass through a lane-private LDS slot: fully unrolled, the 64 evaluations
// of a tile and the accumulator they fill were allocated hundreds of spilled registers (scratch traffic was a quarter
// of this kernel's HBM bytes) and 3 k instructions of code per block row.  Adds the tile's share of the mean.
template <int D, int FAM>
__device__ __forceinline__ void eval_tile(WaveTile<4, NPJW> &acc, const real *pt, kvec_t *mine, real *pk, const pmk_kernel_desc &th,
                                          int row0, int n, int lane)
{
    // the lane's 2 NPJW query points and running means live in its LDS park between block rows (see the kernel)
    constexpr int NC = 2 * NPJW;
    real q[NC][D], mu[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int d = 0; d < D; ++d) q[c][d] = pk[(c * D + d) * PRED_THREADS];
        mu[c] = pk[(NC * D + c) * PRED_THREADS];
    }
#pragma unroll
    for (int pi = 0; pi < 4; ++pi) {
#pragma clang loop unroll_count(2)
        for (int sl = 0; sl < 8; ++sl) {
            const int r = 32 * pi + 2 * frag_irow(lane >> 4, sl >> 1) + (sl & 1);     // row within the block row
            real xr[D];
#pragma unroll
            for (int d = 0; d < D; ++d) xr[d] = pt[d * TILE + r];
            const real cw = pt[D * TILE + r];
            // padding rows carry coordinates of 1e300 (pack_soa): a compactly supported profile is
            // exactly 0 there, so the Spline34 instantiation needs no bounds test
            const bool inside = (FAM == PMK_SPLINE34) || row0 + r < n;
            kvec_t kv;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                kv[c] = inside ? kern_eval<D, FAM, real>(th, q[c], xr) : (real)0;
                mu[c] += kv[c] * cw;
            }
            mine[sl * 64] = kv;
        }
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) {
            const kvec_t kv = mine[sl * 64];
#pragma unroll
            for (int c = 0; c < NC; ++c) acc.f[2 * pi + (sl & 1)][c][sl >> 1] = kv[c];
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) pk[(NC * D + c) * PRED_THREADS] = mu[c];
}

// Schedule of a strip.  One barrier per block row.  Everything that is not an MFMA is kept off the critical path:
//  * the block row's training points are staged into a three-deep LDS ring two block rows ahead;
//  * the TRSM operands are prefetched from the factor into registers (tri_solve_global);
//  * of the two waves that share a SIMD the older one (waves 0..3) gets most of the matrix pipe and finishes its
//    block row ~100 us before the younger one (waves 4..7).  The older wave evaluates its NEXT kernel tile in that
//    idle time, before the barrier; the younger wave evaluates its tile after the barrier, while the older wave --
//    which has the pipe anyway -- is already in its GEMM.  Both evaluations run in the shadow of the other wave's MFMAs;
//  * the lock-step rendezvous with the other strips of the region is split-phase: thread 0 (wave 0, an early
//    finisher) arrives for the next block row and spins, bounded, before the barrier.
template <int D, int FAM>
__global__ __launch_bounds__(PRED_THREADS, NPJW == 1 ? 2 : 1) void predict_strip_kernel(const PatchDesc *__restrict__ descs,
                                                               const real *__restrict__ x, const real *__restrict__ A,
                                                               const real *__restrict__ inv, const real *__restrict__ cvec,
                                                               const StripTask *__restrict__ tasks, int ntasks,
                                                               const int32_t *__restrict__ sorted_item,
                                                               const int32_t *__restrict__ item_query,
                                                               const double *__restrict__ xq, real *__restrict__ strips,
                                                               int64_t strip_stride, pmk_kernel_desc th,
                                                               uint32_t *__restrict__ sync_cnt, int round_base, double min_v,
                                                               double *__restrict__ u_out, double *__restrict__ v_out,
                                                               unsigned long long *__restrict__ clk,
                                                               const double *__restrict__ qdiag)
{
    // the wave number as a scalar: everything decided per wave (is the wave active, which block-row variant) is then a
    // scalar branch, and the buffer resources of the GEMM stay in scalar registers
#ifdef PMK_PRED_SWAVE
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#else
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#endif
    // shader-clock probe: cycles and 10 ns ticks of the lifetime of workgroups 0..7, one per XCD (pmk_ctx_shader_clock)
    unsigned long long c0 = 0, r0 = 0;
    if (clk && blockIdx.x < 8 && threadIdx.x == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    real *V = strips + (int64_t)blockIdx.x * strip_stride + WCOLS * wave;   // this wave's columns, ld = TQ
    constexpr int PTS = (MAX_D + 1) * TILE;
    // training points (SoA) and weights of consecutive block rows: a ring of three with one barrier per block row, of six
    // with one barrier every second block row (PMK_PRED_SYNC2: the older wave of a SIMD then goes straight on into its next
    // GEMM on odd rows instead of waiting for the younger one's block substitution)
#ifdef PMK_PRED_SYNC2
    constexpr int RING = 6;
#else
    constexpr int RING = 3;
#endif
    __shared__ real pts[RING * PTS];
    __shared__ real priv[PRED_WAVES * 8 * 64 * 2 * NPJW];   // lane-private staging of the kernel evaluations (8 slots a lane)
    kvec_t *mine = reinterpret_cast<kvec_t *>(priv) + (wave * 8) * 64 + lane;
    // Per-thread values that are only needed between the MFMA phases (the two query points, the running means and
    // squared norms) are PARKED in LDS: around its GEMM the kernel has no register to spare (128 accumulator + 80 ring
    // registers of 256), and what the compiler spills instead goes to scratch memory, in the middle of the hot loops.
    constexpr int NC = 2 * NPJW;                   // query columns of a lane
    __shared__ real park[(NC * MAX_D + 2 * NC) * PRED_THREADS];
    real *pk = park + threadIdx.x;      // slot k of this thread: pk[k * PRED_THREADS]

    // XCD-aware task order: the workgroups of one XCD take consecutive tasks (= strips of the same
    // region, which stream the same factor L) so that L is fetched into one L2 once per region
    const int nround = (ntasks + (int)gridDim.x - 1) / (int)gridDim.x;
    bool round_sync = true;       // thread 0 only: dropped for good after one missed round barrier
    for (int round = 0; round < nround; ++round) {
        const int base = round * (int)gridDim.x;
        const int nin = min((int)gridDim.x, ntasks - base);          // tasks of this round
        if ((int)blockIdx.x >= nin) break;
        if (round > 0 && threadIdx.x == 0) {
            // round barrier of this XCD's workgroups (speed only, bounded): the strips of a round start together, so
            // that the per-block-row rendezvous below only has to absorb small drifts.  Only the workgroups that have a task
            // in this round arrive (the last round is usually partial: the others have left the loop above).
            uint32_t *cnt = sync_cnt + round_base + 8 * (round - 1) + (blockIdx.x & 7);
            const uint32_t want = ((uint32_t)nin + 7u - (blockIdx.x & 7)) >> 3;   // participants with this id mod 8
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (round_sync) {
                const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
                bool ok;
                while (!(ok = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) &&
                       __builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)PMK_ROUND_TICKS)
                    __builtin_amdgcn_s_sleep(32);
                round_sync = ok;
            }
        }
        const int task = base + xcd_remap(blockIdx.x, nin);
        const StripTask tk = tasks[task];
        const bool active = WCOLS * wave < tk.count;     // wave-uniform
        const PatchDesc pd = descs[tk.region];
        const real *S = A + pd.aoff;
        const real *xs = x + pd.xoff;
        const real *cr = cvec + pd.yoff;
        const int64_t ld = pd.ld;

        // lock-step rendezvous of the strips that share this factor on this XCD (speed only: a bounded spin on an
        // arrival counter, no data is handed over): thread 0 arrives for block row `row` and waits for the others
        auto rendezvous = [&](int row) {
            uint32_t *cnt = sync_cnt + tk.group;
            const uint32_t want = (uint32_t)tk.gsize * (uint32_t)(row + 1);
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want &&
                   __builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)PMK_SYNC_TICKS)
                __builtin_amdgcn_s_sleep(8);
#ifdef PMK_TRACE
            const unsigned long long w = __builtin_amdgcn_s_memrealtime() - t0;
            atomicAdd(&g_sync_stats[0], 1ull);
            if (w >= (unsigned long long)PMK_SYNC_TICKS) atomicAdd(&g_sync_stats[1], 1ull);
            atomicAdd(&g_sync_stats[2], w);
            atomicMax(&g_sync_stats[3], w);
#endif
        };
        // the training points and weights of block row `br`, the same for every lane of the workgroup, by 128 threads
        auto stage_points = [&](int br, int t) {
            real *dst = pts + (br % RING) * PTS;
            const int row = br * TILE + t;
#pragma unroll
            for (int d = 0; d < D; ++d) dst[d * TILE + t] = xs[(int64_t)d * ld + row];
            dst[D * TILE + t] = cr[row];
        };

        // the lane's query columns: 32 pj + 2 (lane & 15) + ej of the wave's WCOLS (padding columns repeat a valid item)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = WCOLS * wave + 32 * (c >> 1) + 2 * (lane & 15) + (c & 1);
            const int64_t p = tk.first + (col < tk.count ? col : 0);
            const int64_t qi = item_query[sorted_item[p]];
#pragma unroll
            for (int d = 0; d < D; ++d) pk[(c * D + d) * PRED_THREADS] = (real)xq[qi * D + d];
            pk[(NC * D + c) * PRED_THREADS] = (real)0;            // mean
            pk[(NC * D + NC + c) * PRED_THREADS] = (real)0;       // |V|^2
        }
        // rows of the last block row that are not identity padding, in 32-row pairs (n = 2000: 3 of 4)
        const int last_pairs = (pd.n - (pd.nt - 1) * TILE + 31) >> 5;

        __syncthreads();                              // every wave is done with the previous task's points
#ifdef PMK_PRED_SYNC2
        if ((int)(threadIdx.x >> 7) < pd.nt) stage_points(threadIdx.x >> 7, threadIdx.x & (TILE - 1));       // rows 0..3
#else
        if (threadIdx.x < 2 * TILE && (int)(threadIdx.x >> 7) < pd.nt) stage_points(threadIdx.x >> 7, threadIdx.x & (TILE - 1));
#endif
        if (threadIdx.x == 0 && tk.group >= 0) rendezvous(0);
        WaveTile<4, NPJW> acc;
        bool have = false;                            // acc already holds this block row's kernel tile
        for (int i = 0; i < pd.nt; ++i) {
#ifdef PMK_PRED_SYNC2
            // even rows only: rows i .. i + 3 were staged before the barrier before this one; rows i + 4, i + 5 go into the
            // slots of rows i - 2, i - 1, which every wave has left behind
            if ((i & 1) == 0) {
                __syncthreads();
                if (threadIdx.x < 2 * TILE && i + 4 + (int)(threadIdx.x >> 7) < pd.nt)
                    stage_points(i + 4 + (int)(threadIdx.x >> 7), threadIdx.x & (TILE - 1));
            }
            PMK_PSTAMP(0);
#else
            __syncthreads();                          // block row i - 1 is complete in every wave; points of row i, i + 1 visible
            PMK_PSTAMP(0);
            if (threadIdx.x < TILE && i + 2 < pd.nt) stage_points(i + 2, threadIdx.x);
#endif
            if (active) {
                if (!have) eval_tile<D, FAM>(acc, pts + (i % RING) * PTS, mine, pk, th, i * TILE, pd.n, lane);
                PMK_PSTAMP(2);
#ifdef PMK_TRACE
                if (round == PMK_TRACE_ROUND && i == PMK_TRACE_ROW && lane == 0 && blockIdx.x < 64)
                    g_row_stamp[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + 1] = __builtin_amdgcn_s_memtime();
#endif
                // ---- acc -= L[i, 0:i] V_0:i  (the strip holds -V), then acc <- -V_i = -L[ii]^-1 acc (block substitution;
                //      the strip wants -V anyway).  Padding rows of the last block row are zero and stay zero: skip them.
                const real *Li = S + (int64_t)i * TILE;
                const real *Lii = Li + (int64_t)i * TILE * ld;
                const real *ninv_i = inv + pd.ioff + (int64_t)i * 4096;
#ifdef PMK_PRED_ONE_VARIANT
                // experiment: one GEMM call site (the last block row computes its identity padding along)
                strip_block_row<4>(acc, Li, ld, V, i, Lii, ninv_i, lane, PMK_TRACED);
#else
                if (i + 1 == pd.nt && last_pairs == 3) strip_block_row<3>(acc, Li, ld, V, i, Lii, ninv_i, lane, PMK_TRACED);
                else if (i + 1 == pd.nt && last_pairs == 2) strip_block_row<2>(acc, Li, ld, V, i, Lii, ninv_i, lane, PMK_TRACED);
                else if (i + 1 == pd.nt && last_pairs == 1) strip_block_row<1>(acc, Li, ld, V, i, Lii, ninv_i, lane, PMK_TRACED);
                else strip_block_row<4>(acc, Li, ld, V, i, Lii, ninv_i, lane, PMK_TRACED);
#endif
                PMK_PSTAMP(4);
                {
                    real vs[NC];
#pragma unroll
                    for (int c = 0; c < NC; ++c) vs[c] = pk[(NC * D + NC + c) * PRED_THREADS];
#pragma unroll
                    for (int fi = 0; fi < 8; ++fi)
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                            for (int c = 0; c < NC; ++c) vs[c] += acc.f[fi][c][qq] * acc.f[fi][c][qq];
#pragma unroll
                    for (int c = 0; c < NC; ++c) pk[(NC * D + NC + c) * PRED_THREADS] = vs[c];
                }
                have = false;
                if (i + 1 < pd.nt) {
                    int srow = tile_i(0, lane, 0);
                    asm volatile("" : "+v"(srow));
#pragma unroll
                    for (int fi = 0; fi < 8; ++fi)
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) {
                            const int row = i * TILE + srow + (tile_i(fi, 0, qq) - tile_i(0, 0, 0));
#pragma unroll
                            for (int pj = 0; pj < NPJW; ++pj) {
                                real2_t o;
                                o[0] = acc.f[fi][2 * pj][qq];
                                o[1] = acc.f[fi][2 * pj + 1][qq];
                                __builtin_nontemporal_store(o, reinterpret_cast<real2_t *>(V + (int64_t)row * TQ + 32 * pj + 2 * (lane & 15)));
                            }
                        }
                    PMK_PSTAMP(5);
                    if (NPJW == 1 && wave < PRED_WAVES / 2) {      // the older wave of its SIMD: next tile now, in its idle time
                        eval_tile<D, FAM>(acc, pts + ((i + 1) % RING) * PTS, mine, pk, th, (i + 1) * TILE, pd.n, lane);
                        have = true;
                    }
                }
                PMK_PSTAMP(6);
            }
            if (threadIdx.x == 0 && tk.group >= 0 && i + 1 < pd.nt) rendezvous(i + 1);
        }
        // ---- reduce over the four lane groups that share a column, then write (u, v)
#pragma unroll
        for (int c = 0; c < NC && active; ++c) {
            real a = pk[(NC * D + c) * PRED_THREADS], b = pk[(NC * D + NC + c) * PRED_THREADS];
            a += __shfl_xor(a, 16); b += __shfl_xor(b, 16);
            a += __shfl_xor(a, 32); b += __shfl_xor(b, 32);
            const int col = WCOLS * wave + 32 * (c >> 1) + 2 * (lane & 15) + (c & 1);
            if ((lane >> 4) == 0 && col < tk.count) {
                real qe[D];
#pragma unroll
                for (int d = 0; d < D; ++d) qe[d] = pk[(c * D + d) * PRED_THREADS];
                const real kself = kern_eval<D, FAM, real>(th, qe, qe);
                double var = (double)kself - (double)b;               // mixtureGP.jl:312
                // the kernel's own diagonal term at the query point (pmk_query_set_diag: DPP kernels, kernel.jl:74, 108)
                if (qdiag) var = ((double)kself + qdiag[item_query[sorted_item[tk.first + col]]]) - (double)b;
                var = var < min_v ? min_v : var;                     // clamp(..., min_v, Inf)
                u_out[tk.first + col] = (double)a;
                v_out[tk.first + col] = var;
            }
        }
        // the next task reuses the strip: order its first stores after this task's last loads
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
    if (clk && blockIdx.x < 8 && threadIdx.x == 0) {
        clk[130 * blockIdx.x + 128] = __builtin_amdgcn_s_memtime() - c0;
        clk[130 * blockIdx.x + 129] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

#if defined(PMK_TRACE) && !defined(PMK_REAL_F32)
extern "C" int pmk_trace_sync_stats(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sync_stats), sizeof(unsigned long long) * 4) != hipSuccess) return -1;
    if (reset == 2) return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_row_stamp), sizeof(unsigned long long) * 512 * 40) == hipSuccess ? 0 : -1;
    if (reset) {
        const unsigned long long z[4] = {0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_sync_stats), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

// the XCD that xcd_remap() gives logical position p of a round of nin workgroups
static int xcd_of_position(int p, int nin)
{
    const int q = nin >> 3, r = nin & 7;
    const int big = r * (q + 1);
    return p < big ? p / (q + 1) : r + (q ? (p - big) / q : 0);
}

// strip tasks for the regions this model owns (host side, after the plan's region offsets are known)
int build_strip_tasks(pmk_query *q, hipStream_t s)
{
    pmk_model *m = q->m;
    std::vector<StripTask> tasks;
    for (int64_t r = 0; r < m->P; ++r) {
        const int64_t b = q->roff[m->leaf_base + r], e = q->roff[m->leaf_base + r + 1];
        if (e <= b) continue;
        // the region's items are dealt evenly over its strips (not TQ, TQ, ..., remainder): strips of one region then
        // take the same time, which is what keeps them in lock-step
#ifdef PMK_HALF_STRIPS      /* diagnostic: only waves 0-3 get columns -> one active wave per SIMD */
        const int64_t nstrips = (e - b + 127) / 128, w = (e - b + nstrips - 1) / nstrips;
#else
        const int64_t nstrips = (e - b + TQ - 1) / TQ, w = (e - b + nstrips - 1) / nstrips;
#endif
        for (int64_t f = b; f < e; f += w) {
            StripTask t;
            t.region = (int32_t)r;
            t.count = (int32_t)((e - f) < w ? (e - f) : w);
            t.first = f;
            t.group = -1;
            t.gsize = 1;
            tasks.push_back(t);
        }
    }
    q->ntasks = (int64_t)tasks.size();
    q->nsync = 0;
    if (tasks.empty()) return 0;
    const int64_t slots = std::min<int64_t>(q->ntasks, (int64_t)m->ctx->num_cu);     // one 8-wave workgroup per CU
    if (std::getenv("PMK_PRED_DEBUG")) {
        int64_t cols = 0;
        for (const StripTask &t : tasks) cols += t.count;
        std::fprintf(stderr, "strips %lld (%.2f rounds of %lld), %lld items = %.1f %% of their columns, last round %lld strips\n",
                     (long long)q->ntasks, (double)q->ntasks / (double)slots, (long long)slots, (long long)cols,
                     100.0 * (double)cols / ((double)q->ntasks * TQ), (long long)(q->ntasks - (q->ntasks - 1) / slots * slots));
    }
    // lock-step groups: consecutive tasks of one round that land on one XCD and stream the same factor
    {
        int64_t g0 = 0;
        auto key = [&](int64_t t) {
            const int64_t round = t / slots, p = t - round * slots;
            const int nin = (int)std::min<int64_t>(slots, q->ntasks - round * slots);
            return std::make_tuple(round, (int64_t)xcd_of_position((int)p, nin), (int64_t)tasks[(size_t)t].region);
        };
        for (int64_t t = 1; t <= q->ntasks; ++t) {
            if (t == q->ntasks || key(t) != key(g0)) {
                if (t - g0 > 1) {
                    for (int64_t u = g0; u < t; ++u) {
                        tasks[(size_t)u].group = (int32_t)q->nsync;
                        tasks[(size_t)u].gsize = (int32_t)(t - g0);
                    }
                    ++q->nsync;
                }
                g0 = t;
            }
        }
    }
    if (q->tasks_cap < q->ntasks) {          // grow only: repeated plans of one query batch reuse the buffer
        if (q->d_tasks) { PMK_HIP(hipFree(q->d_tasks)); q->d_tasks = nullptr; }
        q->tasks_cap = 0;
        const int64_t cap = q->ntasks + q->ntasks / 8 + 64;
        PMK_HIP(hipMalloc(&q->d_tasks, sizeof(StripTask) * (size_t)cap));
        q->tasks_cap = cap;
    }
    // arrival counters: one per lock-step group, then one per (round, XCD) for the round barrier
    q->round_base = q->nsync;
    q->nsync += 8 * ((q->ntasks + slots - 1) / slots);
    if (q->sync_cap < q->nsync) {
        if (q->d_sync) { PMK_HIP(hipFree(q->d_sync)); q->d_sync = nullptr; }
        q->sync_cap = 0;
        const int64_t cap = q->nsync + q->nsync / 8 + 64;
        PMK_HIP(hipMalloc((void **)&q->d_sync, sizeof(uint32_t) * (size_t)cap));
        q->sync_cap = cap;
    }
    PMK_HIP(hipMemcpyAsync(q->d_tasks, tasks.data(), sizeof(StripTask) * tasks.size(), hipMemcpyHostToDevice, s));
    PMK_HIP(hipStreamSynchronize(s));        // `tasks` is a local
    const int64_t stride = (int64_t)m->max_nt * TILE * TQ;
    if (m->strip_slots < slots) {
        if (m->d_strip) PMK_HIP(hipFree(m->d_strip));
        m->d_strip = nullptr;
        m->strip_slots = 0;
        PMK_HIP(hipMalloc(&m->d_strip, sizeof(real) * stride * slots));
        m->strip_slots = slots;
    }
    q->strip_grid = slots;
    return 0;
}

int launch_items(pmk_query *q, const pmk_kernel_desc &th, hipStream_t s)
{
    pmk_model *m = q->m;
    if (q->ntasks == 0) return 0;
    const int64_t stride = (int64_t)m->max_nt * TILE * TQ;
    const StripTask *d_tasks = reinterpret_cast<const StripTask *>(q->d_tasks);
    if (q->nsync > 0) PMK_HIP(hipMemsetAsync(q->d_sync, 0, sizeof(uint32_t) * (size_t)q->nsync, s));
    const bool s34 = th.family == PMK_SPLINE34;
    switch (m->D) {
#define PMK_CASE(DD)                                                                                                   \
    case DD:                                                                                                           \
        if (s34)                                                                                                       \
            hipLaunchKernelGGL((predict_strip_kernel<DD, PMK_SPLINE34>), dim3((unsigned)q->strip_grid), dim3(PRED_THREADS), 0, s, \
                               m->d_desc, (real *)m->d_x, (real *)m->d_a, (real *)m->d_inv, (real *)m->d_c, d_tasks, (int)q->ntasks, q->d_sorted_item,  \
                               q->d_item_query, q->d_xq, (real *)m->d_strip, stride, th, q->d_sync, (int)q->round_base, q->min_v, q->d_u, q->d_v, m->ctx->d_clk, q->d_qdiag);  \
        else                                                                                                           \
            hipLaunchKernelGGL((predict_strip_kernel<DD, 0>), dim3((unsigned)q->strip_grid), dim3(PRED_THREADS), 0, s,  \
                               m->d_desc, (real *)m->d_x, (real *)m->d_a, (real *)m->d_inv, (real *)m->d_c, d_tasks, (int)q->ntasks, q->d_sorted_item,  \
                               q->d_item_query, q->d_xq, (real *)m->d_strip, stride, th, q->d_sync, (int)q->round_base, q->min_v, q->d_u, q->d_v, m->ctx->d_clk, q->d_qdiag);  \
        break;
        PMK_CASE(1) PMK_CASE(2) PMK_CASE(3) PMK_CASE(4)
#undef PMK_CASE
    default:
        set_error("prediction supports input dimension 1..4, got %d", m->D);
        return -2;
    }
    PMK_HIP(hipGetLastError());
    return 0;
}

}  // namespace PMK_NS
}  // namespace pmk
