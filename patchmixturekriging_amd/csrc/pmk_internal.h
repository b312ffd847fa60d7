// Internal declarations shared by the host side and the HIP kernels of libpmk_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/pmk.h"

namespace pmk {

constexpr int TILE = 128;       // factorisation tile edge; slabs are padded to a multiple of it
constexpr int MAX_D = 4;        // input dimension limit (the reference's examples use 1, 2 and 3)
constexpr int TQ = 256;         // query columns per prediction strip (8 waves x 32)

void set_error(const char *fmt, ...);

#define PMK_HIP(expr)                                                                        \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess) {                                                             \
            pmk::set_error("%s failed at %s:%d: %s", #expr, __FILE__, __LINE__,              \
                           hipGetErrorString(e__));                                          \
            return -100;                                                                     \
        }                                                                                    \
    } while (0)

// Per-patch geometry, resident on the device (one entry per patch).
struct PatchDesc {
    int32_t n;        // points in the patch
    int32_t nt;       // ceil(n / TILE)
    int32_t ld;       // nt * TILE : leading dimension of the slab, padded with identity
    int32_t pad_;
    int64_t aoff;     // element offset of the ld x ld slab (K, then L in place)
    int64_t xoff;     // element offset of the SoA coordinates: x[d][ld]
    int64_t yoff;     // element offset into y / z / c (ld each)
    int64_t ioff;     // element offset of the negated inverted 32 x 32 diagonal blocks: nt x 4 x (32 x 32)
};

// A BSP tree in heap order (root 0, children 2i+1 / 2i+2); leaves numbered left to right.
struct BspArrays {
    int D = 0, levels = 0;
    int dot_mode = 0;               // 0: v . x as separate multiplies and adds, 1: as a chain of fused multiply-adds
    int64_t P = 0, N = 0;
    std::vector<double> v;          // (P-1) x D heap order
    std::vector<double> c;          // P-1
    std::vector<int64_t> pre;       // pre-order rank -> heap index
    std::vector<int64_t> leaf_off;  // P+1
    std::vector<int64_t> leaf_inds; // N
};

}  // namespace pmk

struct pmk_bsp {
    pmk::BspArrays t;
};

struct pmk_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    int timers = 0;                 // 0 off, 1 per stage, 2 also per panel launch
    struct Timer { std::string name; hipEvent_t a, b; bool valid; };
    std::vector<Timer> tm;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> panel_ev;   // one pair per panel launch of the last fit
    int panel_n = 0;
    int num_cu = 0;                 // compute units of the device (sizes the persistent prediction grid)
    // shader-clock probe: workgroups 0..7 (one per XCD) of every factorisation step launch and of the prediction strip
    // kernel leave (shader cycles, 100 MHz ticks) of their own lifetime here: block x of 130 pairs for XCD slot x:
    // [2 l], [2 l + 1] for step launch l < 64, [128], [129] for the strip kernel.  The roofline's peak assumes the
    // nominal clock; this says what the kernels actually got (every XCD has its own clock).
    unsigned long long *d_clk = nullptr;
    // pipelined kernel-matrix build: K1 runs block column by block column on a low-priority side stream while the
    // factorisation's step launches (which wait on the per-column events) keep the matrix pipes busy
    hipStream_t side_stream = nullptr;
    hipEvent_t fit_begin = nullptr;
    std::vector<hipEvent_t> col_ev;
    int pipeline_k1 = 0;            // pmk_ctx_set_pipeline(ctx, 1) / PMK_PIPELINE_K1=1 turns it on
    void tic(const char *name);
    void toc(const char *name);
};

struct pmk_model {
    pmk_ctx *ctx = nullptr;
    int D = 0;
    int64_t P = 0;
    int max_nt = 0;
    std::vector<pmk::PatchDesc> desc;   // host copy
    pmk::PatchDesc *d_desc = nullptr;
    int dtype = PMK_F64;                // arithmetic type of the device path (element type of the buffers below)
    size_t esz = 8;
    void *d_x = nullptr;                // SoA coordinates
    void *d_y = nullptr;                // targets (padded with 0)
    void *d_diag = nullptr;             // per-point addend of the kernel's diagonal (pmk_model_set_diag), or null
    void *d_z = nullptr;                // L^-1 y
    void *d_c = nullptr;                // weights
    void *d_a = nullptr;                // slabs
    void *d_inv = nullptr;              // -(L[ss])^-1 for every 32 x 32 diagonal block of L
    int32_t *d_info = nullptr;          // per patch
    // factorisation schedule: patches sorted by tile count (largest first) and, for every threshold t, how many
    // patches have nt >= t (the active prefix of `order` at launch max_nt - t of the end-aligned schedule)
    int32_t *d_order = nullptr;
    std::vector<int32_t> active_prefix;
    // split path (few, large patches: the deep products of a step are cut along K over several workgroups and the two
    // triangular solves run block by block over many workgroups): chosen at creation from P and the tile counts
    bool split_mode = false;
    bool fuse_k1 = false;               // this fit evaluates the strictly lower kernel-matrix tiles inside the factorisation
    // task-queue form of the batched factorisation (one launch for all block columns, pmk_chol.hip): per-XCD task lists
    // and the scheduling block (list heads, error word, per-patch dependency flags), built at the first fit
    std::vector<int32_t> order;         // host copy of d_order
    int queue_mode = 0;                 // 1: task-queue factorisation (PMK_CHOL_QUEUE=1; measured slower than one launch per block column)
    bool queue_built = false, queue_used = false;
    void *d_qtasks = nullptr;
    int qsegs = 1;                      // launches the lists are cut into (1 unless PMK_QUEUE_SEGS says otherwise)
    std::vector<int32_t> qoff;          // [segment][9]: list of XCD x in a segment = tasks[qoff[x] .. qoff[x + 1])
    int32_t *d_sched = nullptr, *d_sched_init = nullptr; size_t sched_bytes = 0;
    int queue_from = 0, queue_l0 = 0;   // first step the queue takes (PMK_QUEUE_FROM; negative: counted from the end)
    int qfstride = 0;
    void *d_partial = nullptr; size_t partial_bytes = 0;      // partial product tiles of one step
    void *d_solve_part = nullptr; size_t solve_bytes = 0;     // partial matrix-vector products of one solve block
    void *d_chain = nullptr; size_t chain_words = 0;          // solve_chain_kernel: error word, then one flag word per block and direction
    int chain_epoch = 0; bool chain_used = false;             // the flags hold the epoch of the launch that set them
    int chain_mode = -1;                                      // -1: by size, 0: block-by-block solves, 1: chained solves (pmk_test.h)
    int64_t tot_a = 0, tot_x = 0, tot_y = 0, tot_inv = 0;
    bool fitted = false;
    pmk_kernel_desc th{};
    double sigma2 = 0;
    // BSP for prediction (heap order on device)
    int levels = 0;
    int64_t P_global = 0, leaf_base = 0;
    int dot_mode = 0;
    double *d_hv = nullptr, *d_hc = nullptr;   // heap order
    int32_t *d_pre = nullptr;                  // pre-order -> heap
    // prediction strip workspace
    void *d_strip = nullptr;
    int64_t strip_slots = 0;
};

struct pmk_query {
    pmk_model *m = nullptr;
    int64_t Nq = 0;
    int64_t nq_cap = 0;             // capacity of the per-point buffers below (grow only)
    int *d_flag = nullptr;          // device scratch flag (region range check of explicit items)
    double *d_xq = nullptr;         // point-major D x Nq
    double *d_qdiag = nullptr;      // per-query addend of k(xq, xq) (pmk_query_set_diag), or null
    int32_t *d_home = nullptr;      // Nq
    int32_t *d_cnt = nullptr;       // Nq : items per query (neighbours + 1)
    int32_t *d_stage_r = nullptr;   // PLAN_STAGE x nq_cap : first neighbour hits of the count pass (region), see plan_kernel
    double *d_stage_t = nullptr;    //                        ... and their t
    int64_t *d_qoff = nullptr;      // Nq+1
    int64_t total = 0;
    int64_t item_cap = 0;           // capacity of the per-item buffers (reused across plans)
    int32_t *d_item_region = nullptr;   // total, reference order
    double *d_item_t = nullptr;         // total (0 for home)
    int32_t *d_item_query = nullptr;    // total
    int32_t *d_sorted_item = nullptr;   // total: sorted position -> item
    int32_t *d_item_pos = nullptr;      // total: item -> sorted position
    int64_t *d_roff = nullptr;          // P_global+1 (of the tree attached when the query was created: roff_P)
    int64_t roff_P = 0;
    std::vector<int64_t> roff;          // host copy
    double *d_u = nullptr, *d_v = nullptr;   // sorted order
    double *d_w = nullptr;                   // reference order (unnormalised), debug
    double *d_yq = nullptr, *d_vq = nullptr; // Nq
    void *d_tmp = nullptr; size_t tmp_bytes = 0;
    void *d_sort_scratch = nullptr; int64_t sort_cap = 0;
    void *d_tasks = nullptr; int64_t ntasks = 0, tasks_cap = 0, strip_grid = 0;
    uint32_t *d_sync = nullptr; int64_t sync_cap = 0, nsync = 0, round_base = 0;   // arrival counters of the strips' lock-step groups   // prediction strip tasks (owned regions)
    bool planned = false;
    double min_v = 1e-12;           // floor of the predictive variance (queryinner!'s keyword min_v, mixtureGP.jl:296)
};

namespace pmk {

// ---- launchers implemented in the .hip files (all enqueue on `s`) ----
// precision-generic kernels live in namespaces f64 / f32 (the same sources compiled twice)
#define PMK_DECLARE_REAL_LAUNCHERS(NS)                                                                              \
    namespace NS {                                                                                                   \
    int launch_kernel_matrix_slabs(const pmk_model *m, const pmk_kernel_desc &th, double sigma2, hipStream_t s,      \
                                   int64_t p0, int64_t np, int stage);                                               \
    int launch_cholesky(pmk_model *m, hipStream_t s, int64_t p0, int64_t np, const hipEvent_t *col_ev, int n_ev);    \
    int launch_backsolve(pmk_model *m, hipStream_t s, int64_t p0, int64_t np);                                       \
    int launch_ninv_from_slabs(pmk_model *m, hipStream_t s);                                                         \
    int set_device_attributes();                                                                                     \
    int build_strip_tasks(pmk_query *q, hipStream_t s);                                                              \
    int launch_items(pmk_query *q, const pmk_kernel_desc &th, hipStream_t s);                                        \
    }
PMK_DECLARE_REAL_LAUNCHERS(f64)
PMK_DECLARE_REAL_LAUNCHERS(f32)
#define PMK_BY_DTYPE(m, CALL) ((m)->dtype == PMK_F32 ? pmk::f32::CALL : pmk::f64::CALL)

int launch_kernel_matrix_dense(const pmk_kernel_desc &th, int D, int64_t n, const double *d_xs, int64_t ldx,
                               int64_t mcols, const double *d_zs, int64_t ldz, double *d_K, int64_t ldk,
                               bool symmetric, hipStream_t s);
int set_plan_attributes();
int launch_iota(int32_t *d, int64_t n, hipStream_t s);
int query_reserve(pmk_query *q, int64_t Nq);
int query_set_items(pmk_query *q, int64_t n, const double *xq, const int32_t *region);
int launch_plan_count(pmk_query *q, double radius, double delta, hipStream_t s);
int launch_plan_fill(pmk_query *q, double radius, double delta, hipStream_t s);
int launch_sort_items(pmk_query *q, hipStream_t s);
int launch_mix(pmk_query *q, const pmk_kernel_desc &wth, int64_t q0, int64_t q1, hipStream_t s);
int launch_export_requests(pmk_query *q, int64_t first, int64_t n, double *x_out, int32_t *region_out, hipStream_t s);
int launch_export_results(pmk_query *q, double *u_out, double *v_out, hipStream_t s);
int grow_item_buffers(pmk_query *q, int64_t total);
int launch_explicit_items(pmk_query *q, int *d_bad, hipStream_t s);
int launch_query_mean(const pmk_kernel_desc *th, int nth, int D, int64_t n, const double *d_xs, int64_t ldx,
                      const double *d_c, int64_t nq, const double *d_xq, double *d_yq, hipStream_t s);
int64_t exclusive_scan_i32_to_i64(const int32_t *d_in, int64_t *d_out, int64_t n, void **tmp, size_t *tmp_bytes,
                                  hipStream_t s);

}  // namespace pmk
