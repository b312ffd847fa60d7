// K4: per-(query, region) prediction, queryinner! of the reference (src/RKHS/mixtureGP.jl:296-316)
// batched over all the queries that touch a region:
//     kq = k(xq, X_r)            (cross-kernel tile, evaluated in registers, never written to HBM)
//     u  = kq . c_r              (mean)
//     V  = L_r^-1 kq             (blocked forward substitution, every product on fp64 MFMA)
//     v  = clamp(k(xq,xq) - |V|^2, 1e-12, inf)
// The reference runs one dtrsv per (query, region) and streams the 4 n^2-byte factor each time; here a
// workgroup owns a strip of TQ = 128 query columns (4 waves x 32 columns, waves independent) and
// sweeps the block rows of L once for all of them.
//
// Strip task, block row i:  acc(128 x 32) = Kq_i - sum_{j<i} L[i,j] V_j   (gemm_nt, V_j re-read from the
// wave's strip in global memory, which is stored negated so the MFMA accumulates the subtraction)
//                           -V_i = -L[ii]^-1 acc                          (tri_solve_inplace: block substitution)
#include <cstdlib>

#include "pmk_mfma.h"

namespace pmk {
namespace PMK_NS {

constexpr int PF_PRED = 4;
#ifndef PMK_PFJ
#define PMK_PFJ 4
#endif
constexpr int PFJ_PRED = PMK_PFJ;   // J-operand (the wave's own strip columns, HBM) prefetch depth

struct StripTask {
    int32_t region;    // local patch index in the model
    int32_t count;     // valid columns (<= TQ)
    int64_t first;     // first sorted item of the strip
};

template <int D, int FAM>
__global__ __launch_bounds__(256, 2) void predict_strip_kernel(const PatchDesc *__restrict__ descs,
                                                               const real *__restrict__ x, const real *__restrict__ A,
                                                               const real *__restrict__ inv, const real *__restrict__ cvec,
                                                               const StripTask *__restrict__ tasks, int ntasks,
                                                               const int32_t *__restrict__ sorted_item,
                                                               const int32_t *__restrict__ item_query,
                                                               const double *__restrict__ xq, real *__restrict__ strips,
                                                               int64_t strip_stride, pmk_kernel_desc th,
                                                               double *__restrict__ u_out, double *__restrict__ v_out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    real *V = strips + (int64_t)blockIdx.x * strip_stride + 32 * wave;   // this wave's 32 columns, ld = TQ
    __shared__ real tri[TRI_LDS_DOUBLES];     // TRSM operands of the current block row (shared by the 4 waves)

    // XCD-aware task order: the workgroups of one XCD take consecutive tasks (= strips of the same
    // region, which stream the same factor L) so that L is fetched into one L2 once per region
    const int nround = (ntasks + (int)gridDim.x - 1) / (int)gridDim.x;
    for (int round = 0; round < nround; ++round) {
        const int base = round * (int)gridDim.x;
        const int nin = min((int)gridDim.x, ntasks - base);          // tasks of this round
        if ((int)blockIdx.x >= nin) break;
        const int task = base + xcd_remap(blockIdx.x, nin);
        const StripTask tk = tasks[task];
        const bool active = 32 * wave < tk.count;     // wave-uniform; idle waves still help stage operands
        const PatchDesc pd = descs[tk.region];
        const real *S = A + pd.aoff;
        const real *xs = x + pd.xoff;
        const real *cr = cvec + pd.yoff;
        const int64_t ld = pd.ld;

        // the lane's 2 query columns: 2 (lane & 15) + ej of the wave's 32
        real q[2][D];
        int64_t pos[2];
        bool valid[2];
#pragma unroll
        for (int ej = 0; ej < 2; ++ej) {
            const int col = 32 * wave + 2 * (lane & 15) + ej;
            valid[ej] = col < tk.count;
            pos[ej] = tk.first + (valid[ej] ? col : 0);                // padding columns repeat a valid item
            const int64_t qi = item_query[sorted_item[pos[ej]]];
#pragma unroll
            for (int d = 0; d < D; ++d) q[ej][d] = (real)xq[qi * D + d];
        }
        real mu[2] = {0.0, 0.0}, vs[2] = {0.0, 0.0};

        for (int i = 0; i < pd.nt; ++i) {
            __syncthreads();                          // every wave is done with the previous block row's operands
            stage_tri_operands(tri, S + (int64_t)i * TILE + (int64_t)i * TILE * ld, ld, inv + pd.ioff + (int64_t)i * 4096,
                               threadIdx.x, 256);
            __syncthreads();
            if (!active) continue;
            WaveTile<4, 1> acc;
            // ---- Kq tile for block row i (query is the first kernel argument, mixtureGP.jl:304)
#pragma unroll
            for (int fi = 0; fi < 8; ++fi)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int row = i * TILE + tile_i(fi, lane, qq);
                    real xr[D];
#pragma unroll
                    for (int d = 0; d < D; ++d) xr[d] = xs[(int64_t)d * ld + row];
                    const real cw = cr[row];
                    // padding rows carry coordinates of 1e300 (pack_soa): a compactly supported profile is
                    // exactly 0 there, so the Spline34 instantiation needs no bounds test in its unrolled tile
                    const bool inside = (FAM == PMK_SPLINE34) || row < pd.n;
#pragma unroll
                    for (int ej = 0; ej < 2; ++ej) {
                        const real kv = inside ? kern_eval<D, FAM, real>(th, q[ej], xr) : 0.0;
                        acc.f[fi][ej][qq] = kv;
                        mu[ej] += kv * cw;
                    }
                }
            // ---- acc -= L[i, 0:i] V_0:i   (the strip holds -V)
            if (i > 0) {
                // order this wave's earlier strip stores before its loads of them
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                gemm_nt<4, 1, PF_PRED, PFJ_PRED>(acc, S + (int64_t)i * TILE, ld, V, TQ, i * TILE, lane);
            }
            // ---- acc <- -V_i = -L[ii]^-1 acc   (block substitution; the strip wants -V anyway)
            tri_solve_inplace<1>(acc, tri, lane);
#pragma unroll
            for (int fi = 0; fi < 8; ++fi)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                    for (int ej = 0; ej < 2; ++ej) vs[ej] += acc.f[fi][ej][qq] * acc.f[fi][ej][qq];
            if (i + 1 < pd.nt) {
#pragma unroll
                for (int fi = 0; fi < 8; ++fi)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const int row = i * TILE + tile_i(fi, lane, qq);
                        real2_t o;
                        o[0] = acc.f[fi][0][qq];
                        o[1] = acc.f[fi][1][qq];
                        __builtin_nontemporal_store(o, reinterpret_cast<real2_t *>(V + (int64_t)row * TQ + 2 * (lane & 15)));
                    }
            }
        }
        // ---- reduce over the four lane groups that share a column, then write (u, v)
#pragma unroll
        for (int ej = 0; ej < 2 && active; ++ej) {
            real a = mu[ej], b = vs[ej];
            a += __shfl_xor(a, 16); b += __shfl_xor(b, 16);
            a += __shfl_xor(a, 32); b += __shfl_xor(b, 32);
            if ((lane >> 4) == 0 && valid[ej]) {
                const real kself = kern_eval<D, FAM, real>(th, q[ej], q[ej]);
                double var = (double)kself - (double)b;               // mixtureGP.jl:312
                var = var < 1e-12 ? 1e-12 : var;
                u_out[pos[ej]] = (double)a;
                v_out[pos[ej]] = var;
            }
        }
        // the next task reuses the strip: order its first stores after this task's last loads
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
}

// strip tasks for the regions this model owns (host side, after the plan's region offsets are known)
int build_strip_tasks(pmk_query *q, hipStream_t s)
{
    pmk_model *m = q->m;
    std::vector<StripTask> tasks;
    for (int64_t r = 0; r < m->P; ++r) {
        const int64_t b = q->roff[m->leaf_base + r], e = q->roff[m->leaf_base + r + 1];
        for (int64_t f = b; f < e; f += TQ) {
            StripTask t;
            t.region = (int32_t)r;
            t.count = (int32_t)((e - f) < TQ ? (e - f) : TQ);
            t.first = f;
            tasks.push_back(t);
        }
    }
    q->ntasks = (int64_t)tasks.size();
    if (tasks.empty()) return 0;
    if (q->tasks_cap < q->ntasks) {          // grow only: repeated plans of one query batch reuse the buffer
        if (q->d_tasks) { PMK_HIP(hipFree(q->d_tasks)); q->d_tasks = nullptr; }
        q->tasks_cap = 0;
        const int64_t cap = q->ntasks + q->ntasks / 8 + 64;
        PMK_HIP(hipMalloc(&q->d_tasks, sizeof(StripTask) * (size_t)cap));
        q->tasks_cap = cap;
    }
    PMK_HIP(hipMemcpyAsync(q->d_tasks, tasks.data(), sizeof(StripTask) * tasks.size(), hipMemcpyHostToDevice, s));
    PMK_HIP(hipStreamSynchronize(s));        // `tasks` is a local
    const int64_t slots = std::min<int64_t>(q->ntasks, 2 * (int64_t)m->ctx->num_cu);
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
    const bool s34 = th.family == PMK_SPLINE34;
    switch (m->D) {
#define PMK_CASE(DD)                                                                                                   \
    case DD:                                                                                                           \
        if (s34)                                                                                                       \
            hipLaunchKernelGGL((predict_strip_kernel<DD, PMK_SPLINE34>), dim3((unsigned)q->strip_grid), dim3(256), 0, s, \
                               m->d_desc, (real *)m->d_x, (real *)m->d_a, (real *)m->d_inv, (real *)m->d_c, d_tasks, (int)q->ntasks, q->d_sorted_item,  \
                               q->d_item_query, q->d_xq, (real *)m->d_strip, stride, th, q->d_u, q->d_v);                       \
        else                                                                                                           \
            hipLaunchKernelGGL((predict_strip_kernel<DD, 0>), dim3((unsigned)q->strip_grid), dim3(256), 0, s,           \
                               m->d_desc, (real *)m->d_x, (real *)m->d_a, (real *)m->d_inv, (real *)m->d_c, d_tasks, (int)q->ntasks, q->d_sorted_item,  \
                               q->d_item_query, q->d_xq, (real *)m->d_strip, stride, th, q->d_u, q->d_v);                       \
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
