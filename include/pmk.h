/*
 * pmk.h -- C ABI of libpmk_hip.so: the MI355X (gfx950) implementation of the per-patch
 * GP-regression hot path of PatchMixtureKriging.
 *
 * The reference (pure Julia) has no FFI; the surface this ABI replaces is the set of Julia
 * functions that examples/mixGP.jl and examples/IBB1D.jl call.  Each entry point cites the
 * reference function (file:line relative to the reference tree) whose work it takes over.
 * The Julia binding (julia/PatchMixtureKriging) and the Python mirror
 * (patchmixturekriging_amd) both sit on exactly these symbols; INTEGRATION.md shows the
 * ccall stubs.
 *
 * Conventions
 *   - plain C types only; no exceptions or callbacks cross the boundary
 *   - every function returns int status: 0 ok, <0 bad argument / runtime failure
 *     (text via pmk_last_error()), >0 numerical failure
 *   - points are packed point-major: X[d + D*i]  (the D x N column-major matrix of
 *     array2matrix, src/misc/utilities.jl:25-36)
 *   - dense matrices are column-major; all indices are 0-based (Julia wrappers add 1)
 *   - host pointers unless the name says _dev; the caller owns host buffers, the library
 *     owns device memory behind the opaque handles
 *   - fp64 (the reference is Float64-only: src/RKHS/RKHS.jl:4-11, partition.jl:135)
 *   - one call at a time per context (the reference is single-threaded); calls that return
 *     host data block until it is there, the staged *_run / *_fit calls only enqueue on the
 *     context's stream
 */
#ifndef PMK_H
#define PMK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMK_VERSION 103

/* kernel families = the isbits kernel structs of src/misc/declarations.jl:18-45,65-67,75-111 */
enum {
    PMK_SPLINE34 = 1,  /* Spline34KernelType(a)                    kernel.jl:299-313 */
    PMK_SPLINE12 = 2,  /* Spline12KernelType(a)                    kernel.jl:316-330 */
    PMK_SPLINE32 = 3,  /* Spline32KernelType(a)                    kernel.jl:333-347 */
    PMK_GAUSSIAN = 4,  /* GaussianKernel1DType(eps_sq)             kernel.jl:350-357 */
    PMK_RQ       = 5,  /* RationalQuadraticKernelType(a)           kernel.jl:360-366 */
    PMK_TRQ      = 6,  /* TunableRationalQuadraticKernelType(a,w)  kernel.jl:368-374 */
    PMK_MODSQEXP = 7,  /* ModulatedSqExpKernelType(eps_sq,nu), D=1 kernel.jl:376-391 */
    PMK_BB10     = 10, /* BrownianBridge10                         kernel.jl:156-158 */
    PMK_BB20     = 11, /* BrownianBridge20                         kernel.jl:218-225 */
    PMK_BB1EPS   = 12, /* BrownianBridge1eps(eps)                  kernel.jl:168-174 */
    PMK_BB2EPS   = 13  /* BrownianBridge2eps(eps)                  kernel.jl:176-193 */
};
#define PMK_FLAG_SEMIINF 1 /* BrownianBridgeSemiInfDomain{base}      kernel.jl:256-263 */

typedef struct pmk_kernel_desc {
    int32_t family;
    int32_t flags;
    double  p[4];
} pmk_kernel_desc;

/* arithmetic type of the device path.  The reference is Float64-only (RKHS.jl:4-11): PMK_F64 is the parity
 * path; PMK_F32 (fp32 storage + v_mfma_f32, BASELINE config E) has no exact reference semantics and is judged
 * against the fp64 oracle with eps32-scaled bounds.  Host buffers are double in both cases. */
enum { PMK_F64 = 0, PMK_F32 = 1 };

typedef struct pmk_ctx   pmk_ctx;    /* device + stream + workspaces */
typedef struct pmk_bsp   pmk_bsp;    /* host BSP tree (root of setuppartition) */
typedef struct pmk_model pmk_model;  /* fitted MixtureGPType on the device */
typedef struct pmk_query pmk_query;  /* a resident batch of query points + its work items */

int         pmk_version(void);
const char *pmk_last_error(void);

/* ---- context ------------------------------------------------------------------------- */
int  pmk_ctx_create(int device, pmk_ctx **out);
/* launch on a caller-owned hipStream_t (e.g. torch's current stream); NULL -> context's own (a non-blocking stream:
 * NOT ordered with the device's default stream) */
int  pmk_ctx_set_stream(pmk_ctx *ctx, void *hip_stream);
/* launch on the device's legacy default (null) stream -- the stream a framework's "default stream" is: the handle of that
 * stream is 0, which pmk_ctx_set_stream reads as "the context's own" */
int  pmk_ctx_set_stream_null(pmk_ctx *ctx);
/* on = 1: pmk_model_fit builds the kernel matrix block column by block column on a low-priority side stream while the
 * factorisation's launches run (results are bit-identical either way).  Default 0 = build it first, on the context's
 * stream: on MI355X the overlap measured 0.3 ms slower per 256 x 2000 fit (the step launches leave no room).  Never
 * pipelined while stage timers are on or on the split path.  Environment: PMK_PIPELINE_K1=1. */
int  pmk_ctx_set_pipeline(pmk_ctx *ctx, int on);
int  pmk_ctx_synchronize(pmk_ctx *ctx);
void pmk_ctx_destroy(pmk_ctx *ctx);
/* elapsed ms of the most recent staged call's named stage ("kernel_matrix", "cholesky",
 * "solve", "plan", "items", "mix"); enabled by pmk_ctx_enable_timers(ctx, 1) */
int  pmk_ctx_enable_timers(pmk_ctx *ctx, int on);
int  pmk_ctx_timer_ms(pmk_ctx *ctx, const char *stage, double *ms);
/* shader clock (GHz) that workgroups 0..7 (one per XCD) saw over their lifetime in the factorisation step launches of
 * the last fit (which = 0) or in the last prediction strip kernel (which = 1), time-weighted mean; 0 if none has run.  The fp64 MFMA peak that the
 * rooflines are priced against assumes the nominal 2.4 GHz; under these kernels the chip runs slower. */
int  pmk_ctx_shader_clock(pmk_ctx *ctx, int which, double *ghz);

/* ---- BSP: host, exact (integer outputs are part of the parity contract) --------------- */
/* setuppartition(X, levels)  src/patchwork/partition.jl:106-129 (+ gethyperplane :86-100,
 * splitpoints :64-83, createchildren :166-217, labelleafnodes :131-159).
 * Two Julia-stdlib behaviours that the reference's text does not fix are switches, stored with the tree and honoured by
 * every later search on it (host and device):
 *   sign_mode +1: v = +z/|z| ; -1: v = -sign(z1) z/|z|           (sign convention of svd)
 *   dot_mode   0: dot(v, x) = separate multiplies and adds ; 1: a chain of fused multiply-adds
 *                 (LinearAlgebra.dot -> BLAS ddot: whether its short loop was contracted depends on the BLAS build) */
int     pmk_bsp_build(int D, int64_t N, const double *X, int levels, int sign_mode, int dot_mode, pmk_bsp **out);
/* the same build on the GPU (X: host or device pointer, point-major N x D; N < 2^31): level-by-level segmented
 * pairwise sums, radix-sort medians and stable splits; every output equals pmk_bsp_build's bit for bit.  Blocks. */
int     pmk_bsp_build_device(pmk_ctx *ctx, int D, int64_t N, const double *X, int levels, int sign_mode, int dot_mode,
                             pmk_bsp **out);
/* rebuild a tree from its pre-order hyperplanes (for shipping a tree between processes) */
int     pmk_bsp_from_hyperplanes(int D, int levels, const double *hp_v, const double *hp_c, int dot_mode, pmk_bsp **out);
void    pmk_bsp_destroy(pmk_bsp *bsp);
int     pmk_bsp_dim(const pmk_bsp *bsp);
int     pmk_bsp_levels(const pmk_bsp *bsp);
int     pmk_bsp_dot_mode(const pmk_bsp *bsp);
int64_t pmk_bsp_num_leaves(const pmk_bsp *bsp);
int64_t pmk_bsp_num_points(const pmk_bsp *bsp);
/* fetchhyperplanes(root)  src/RKHS/mixtureGP.jl:322-334 : pre-order; hp_v is D x (P-1).
 * leaf_offsets[P+1], leaf_inds[N]: X_parts_inds of setuppartition (ascending per leaf).
 * Any pointer may be NULL. */
int     pmk_bsp_arrays(const pmk_bsp *bsp, double *hp_v, double *hp_c,
                       int64_t *leaf_offsets, int64_t *leaf_inds);
/* organizetrainingsets(root, levels, X0, eps)  partition.jl:301-357 (+ :269-298).
 * offsets[P+1] always written; inds[offsets[P]] = X_set_inds grouped by region;
 * list_offsets[N+1] + lists = regions_list_set.  inds/list_offsets/lists may be NULL
 * (call once with NULLs to size the buffers). */
int     pmk_bsp_assign(const pmk_bsp *bsp, int64_t N, const double *X, double eps,
                       int64_t *offsets, int64_t *inds, int64_t *list_offsets, int64_t *lists);
/* the same assignment on the GPU (X: host or device pointer; outputs on the host, identical to pmk_bsp_assign's) */
int     pmk_bsp_assign_device(pmk_ctx *ctx, const pmk_bsp *bsp, int64_t N, const double *X, double eps,
                              int64_t *offsets, int64_t *inds, int64_t *list_offsets, int64_t *lists);
/* findpartition(x, root, levels)  partition.jl:248-262 ; returns the leaf or <0 */
int64_t pmk_bsp_findpartition(const pmk_bsp *bsp, const double *x);
/* findneighbourpartitions(p, radius, root, levels, hps, home; delta)  mixtureGP.jl:339-405.
 * returns the number kept (or <0); region_inds[<=P-1]; ts[P-1], zs[D x (P-1)], keep[P-1]
 * may be NULL. */
int64_t pmk_bsp_neighbours(const pmk_bsp *bsp, const double *p, double radius, double delta,
                           int64_t home, int64_t *region_inds, double *ts, double *zs, uint8_t *keep);

/* ---- kernel matrix (device compute, host in/out) -------------------------------------- */
/* constructkernelmatrix(X, theta)  src/RKHS/RKHS.jl:4-34  (Z == NULL: n x n, exactly symmetric)
 * constructkernelmatrix(X, Z, theta)  RKHS.jl:95-110      (Z != NULL: n x m) */
int pmk_kernel_matrix(pmk_ctx *ctx, const pmk_kernel_desc *th, int D,
                      int64_t n, const double *X, int64_t m, const double *Z,
                      double *K, int64_t ldk);

/* ---- fit ------------------------------------------------------------------------------ */
/* MixtureGPType(X_set, hps) + upload: src/RKHS/mixtureGP.jl:54-66.  P patches, patch r has
 * n[r] points X[r] (D x n[r]) and targets y[r].  Inputs become device-resident. */
int  pmk_model_create(pmk_ctx *ctx, int D, int64_t P, const int64_t *n,
                      const double *const *X, const double *const *y, pmk_model **out);
/* same with an explicit arithmetic type (PMK_F64 / PMK_F32) */
int  pmk_model_create_ex(pmk_ctx *ctx, int D, int64_t P, const int64_t *n,
                         const double *const *X, const double *const *y, int dtype, pmk_model **out);
/* fitmixtureGP!(eta, y_parts, theta, sigma2)  mixtureGP.jl:70-118 on the resident inputs:
 * per patch K (RKHS.jl:13-34), U = K + sigma2 I, L = chol(U), c = U^-1 y (one Cholesky
 * serves both; the reference's separate LU of :106 is not repeated).  Enqueues only. */
int  pmk_model_fit(pmk_model *m, const pmk_kernel_desc *th, double sigma2);
/* blocks; info[P]: 0 ok, k>0 leading minor k not positive definite (PosDefException(k)) */
int  pmk_model_info(pmk_model *m, int32_t *info);
/* replace the resident targets (same sizes) */
int  pmk_model_set_targets(pmk_model *m, const double *const *y);
/* Per-point addend of the kernel's DIAGONAL: the next fits use K[i][i] = k(x_i, x_i) + diag[r][i] (+ sigma2).  This is
 * what the reference's AdaptiveKernelDPPType / AdaptiveKernelMultiWarpDPPType add where p == q (src/RKHS/kernel.jl:70-75,
 * 102-110: 1 + g(p)^2, resp. 1 + self_gain sum a_m |w_m(p)|), the rest of those kernels being a stationary kernel on
 * positions + appended warp values.  diag = NULL clears it.  Blocks. */
int  pmk_model_set_diag(pmk_model *m, const double *const *diag);
enum { PMK_GET_C = 0, PMK_GET_L = 1, PMK_GET_K = 2, PMK_GET_LINV_DIAG = 3 };
/* pull c_set[r] (n), L_set[r] (n x n lower, strict upper zero), U_set[r] (n x n, K without
 * noise, rebuilt on demand), or the negated inverses of the 32 x 32 diagonal blocks of L
 * (4 ceil(n/128) blocks, 32 x 32 column-major each; the TRSM operands, for tests) */
int  pmk_model_get(pmk_model *m, int64_t patch, int what, double *out, int64_t ld);
int64_t pmk_model_num_patches(const pmk_model *m);
void pmk_model_destroy(pmk_model *m);
/* one-shot convenience = create + fit + info + c_out (rows 13 and 17 of the scope table;
 * fitRKHS! src/RKHS/RKHS.jl:182-217 is the P == 1 case).  c_out[r] may be NULL. */
int  pmk_fit_batched(pmk_ctx *ctx, const pmk_kernel_desc *th, double sigma2, int D, int64_t P,
                     const int64_t *n, const double *const *X, const double *const *y,
                     pmk_model **out, double *const *c_out, int32_t *info);

/* rebuild a device model from host factors (c_set, L_set of a fitted MixtureGPType; L[r] is n[r] x n[r]
 * column-major with leading dimension ldl[r], lower triangle used): the checkpoint/resume path, and what
 * queryinner(xq, X, theta, c, L) needs.  The TRSM operands are recomputed on the device. */
int  pmk_model_load(pmk_ctx *ctx, int D, int64_t P, const int64_t *n, const double *const *X,
                    const double *const *c, const double *const *L, const int64_t *ldl, pmk_model **out);
/* queryinner(xq, X, theta, c, L)  src/RKHS/mixtureGP.jl:296-320, batched over Nq points against ONE patch:
 * mu[j] = k(xq_j, X).c,  var[j] = clamp(k(xq_j,xq_j) - |L^-1 k(xq_j, X)|^2, 1e-12, inf).  No tree needed. */
int  pmk_model_queryinner(pmk_model *m, int64_t patch, const pmk_kernel_desc *th, int64_t Nq, const double *Xq,
                          double *mu, double *var);
/* the same with the keyword min_v of queryinner! (mixtureGP.jl:296): var = max(k(x,x) - |L^-1 k|^2, min_v).  min_v =
 * -HUGE_VAL gives the unclamped variance term1 - term2 of evalqueryGP! (src/RKHS/querying.jl:61-79) */
int  pmk_model_queryinner_ex(pmk_model *m, int64_t patch, const pmk_kernel_desc *th, int64_t Nq, const double *Xq,
                             double min_v, double *mu, double *var);
/* replace the resident weights c_set (setupGPquery(c, X, theta, sigma2), querying.jl:43-59, takes c from its caller:
 * fit for the factor of K + sigma2 I, then put the caller's c in place) */
int  pmk_model_set_weights(pmk_model *m, const double *const *c);
/* all weight vectors at once: c[r] receives the n_r weights of patch r (one device-to-host transfer for the whole
 * model; what fitmixtureGP! stores into c_set, mixtureGP.jl:106,115) */
int  pmk_model_get_weights(pmk_model *m, double *const *c);

/* ---- predict -------------------------------------------------------------------------- */
/* attach the tree; this model holds the global leaves [leaf_base, leaf_base + P) */
int  pmk_model_set_bsp(pmk_model *m, const pmk_bsp *bsp, int64_t leaf_base);
/* upload Nq query points (Xq: host or device pointer, point-major Nq x D) */
int  pmk_query_create(pmk_model *m, int64_t Nq, const double *Xq, pmk_query **out);
/* stage 1: home leaf (partition.jl:248-262) + neighbour items (mixtureGP.jl:339-405) for
 * every query, items sorted by region (stable).  Blocks (sizes come back to the host). */
int  pmk_query_plan(pmk_query *q, double radius, double delta);
/* number of (query, region) items in total and in this model's regions */
int  pmk_query_counts(pmk_query *q, int64_t *total_items, int64_t *first_owned, int64_t *num_owned);
/* region_offsets[P_global+1] of the sorted item list (host copy) */
int  pmk_query_region_offsets(pmk_query *q, int64_t *region_offsets);
/* stage 2: queryinner! (mixtureGP.jl:296-316) for every owned item: u = kq.c,
 * v = clamp(k(x,x) - |L^-1 kq|^2, 1e-12, inf).  Enqueues only. */
int  pmk_query_items(pmk_query *q, const pmk_kernel_desc *th);
/* device pointers of the per-item results in sorted order (length total_items each): where a multi-GPU caller that
 * drives the exchange itself deposits the (u, v) it gets back from the leaf owners between stage 2 and 3 */
int  pmk_query_item_buffers(pmk_query *q, void **u_dev, void **v_dev);
/* ---- multi-GPU, queries sharded over ranks: a rank plans only its own queries against the
 * global tree, sends the (point, region) requests of every sorted-list segment to the rank that owns those leaves,
 * which evaluates queryinner! for them and sends (u, v) back into the requester's item buffers.
 * requests of the sorted items [first, first + n) -> DEVICE arrays xq_dev [n x D point-major], region_dev [n] */
int  pmk_query_export_requests(pmk_query *q, int64_t first, int64_t n, double *xq_dev, int32_t *region_dev);
/* per-query addend of k(xq, xq) in the predictive variance (the same diagonal term for a query point; NULL clears) */
int  pmk_query_set_diag(pmk_query *q, const double *diag);
/* a planned batch of n explicit (point, region) items, one per point (host or device pointers); every region must
 * lie in this model's leaves (-3 otherwise).  Follow with pmk_query_items + pmk_query_export_results. */
int  pmk_query_create_items(pmk_model *m, int64_t n, const double *xq, const int32_t *region, pmk_query **out);
/* (u, v) of all items in item order (the order given to pmk_query_create_items; reference order for a planned
 * query) -> DEVICE arrays of length total_items (either may be NULL).  Enqueues. */
int  pmk_query_export_results(pmk_query *q, double *u_dev, double *v_dev);
/* ---- multi-GPU inside the library: an RCCL communicator (one rank per GPU, xGMI) owned by a context --------------
 * rank 0 calls pmk_comm_unique_id and ships the 128 bytes to the other ranks by any host channel (MPI.jl, a file,
 * torch.distributed ...); every rank then calls pmk_comm_create (collective).  RCCL is bound at run time. */
#define PMK_COMM_ID_BYTES 128
typedef struct pmk_comm pmk_comm;
int  pmk_comm_unique_id(void *id_out);
int  pmk_comm_create(pmk_ctx *ctx, int rank, int world, const void *id, pmk_comm **out);
int  pmk_comm_rank(const pmk_comm *comm);
int  pmk_comm_size(const pmk_comm *comm);
void pmk_comm_destroy(pmk_comm *comm);
/* (first, count) of every rank's segment of a region-sorted item list (rank r owns leaves [r P/world, (r+1) P/world)) */
int  pmk_shard_segments(const int64_t *region_offsets, int64_t P_global, int world, int64_t *first, int64_t *count);
/* one predict step of a model whose leaves and queries are sharded over the communicator (collective):
 * querymixtureGP! (mixtureGP.jl:159-294) for THIS rank's queries = plan -> requests to the leaf owners (grouped
 * ncclSend/ncclRecv) -> queryinner! for everything received -> (u, v) back -> mixture.  Enqueues on the context's
 * stream after the plan; results with pmk_query_fetch.  total_items (may be NULL): this rank's item count. */
int  pmk_query_predict_sharded(pmk_query *q, pmk_comm *comm, const pmk_kernel_desc *th,
                               const pmk_kernel_desc *weight_th, double radius, double delta, int64_t *total_items);
/* the same step with REPLICATED queries (every rank passes all queries): plan of all queries -> queryinner! for the items
 * in this rank's leaves -> one ncclAllGather of padded (u, v) slices (every rank knows every slice's size from its own
 * plan) -> mixture of all queries on every rank.  BASELINE.json's "RCCL all-gather ... of the per-patch predictions before
 * the mixture weights are applied", literally; collective. */
int  pmk_query_predict_allgather(pmk_query *q, pmk_comm *comm, const pmk_kernel_desc *th,
                                 const pmk_kernel_desc *weight_th, double radius, double delta, int64_t *total_items);
/* payload bytes this rank sent / received in the exchange of its last predict step (either form) */
int  pmk_comm_last_bytes(const pmk_comm *comm, int64_t *sent, int64_t *received);
/* stage 3: mixture weights and blend (mixtureGP.jl:224-272) for queries [q0, q1).  Enqueues. */
int  pmk_query_mix(pmk_query *q, const pmk_kernel_desc *weight_th, int64_t q0, int64_t q1);
/* blocks; Yq, Vq [Nq] (either may be NULL) */
int  pmk_query_fetch(pmk_query *q, double *Yq, double *Vq);
/* debug_vars of MixtureGPDebugType (mixtureGP.jl:5-35): home[Nq], item_offsets[Nq+1],
 * then per item in reference order (neighbours in hyperplane order, home last):
 * item_region, item_t (0 for home), item_w (unnormalised), item_u, item_v.  NULLs allowed. */
int  pmk_query_debug(pmk_query *q, int64_t *home, int64_t *item_offsets, int64_t *item_region,
                     double *item_t, double *item_w, double *item_u, double *item_v);
void pmk_query_destroy(pmk_query *q);
/* one-shot querymixtureGP!(Yq,Vq,Xq,eta,root,levels,radius,delta,theta,sigma2,weight_theta,..)
 * src/RKHS/mixtureGP.jl:159-294 for a model that holds every leaf */
int  pmk_predict_mixture(pmk_model *m, const pmk_kernel_desc *th, const pmk_kernel_desc *weight_th,
                         int64_t Nq, const double *Xq, double radius, double delta,
                         double *Yq, double *Vq);
/* query!(Yq, Xq, eta)  src/RKHS/RKHS.jl:220-247 : mean only, Yq = K(Xq, X) c */
int  pmk_query_mean(pmk_ctx *ctx, const pmk_kernel_desc *th, int D, int64_t n, const double *X,
                    const double *c, int64_t Nq, const double *Xq, double *Yq);
/* query!(Yq, Xq, eta::RKHSProblemType{Vector{KT}})  RKHS.jl:278-305 : one kernel per centre, ths[n] */
int  pmk_query_mean_multi(pmk_ctx *ctx, const pmk_kernel_desc *ths, int D, int64_t n, const double *X,
                          const double *c, int64_t Nq, const double *Xq, double *Yq);

#ifdef __cplusplus
}
#endif
#endif
