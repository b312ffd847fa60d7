/*
 * pmk_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the per-patch GP hot path of the reference
 * (RoyCCWang/PatchMixtureKriging, pure Julia).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path
 * (patchmixturekriging_amd + libpmk_hip.so) never does.
 *
 * PARITY UNPINNED: the reference ships no golden vectors or tests
 * (test/runtests.jl:4-6 is an empty testset) and Julia is not installed in the build
 * container, so the reference cannot be executed.  The oracle is pinned only by
 *   - closed-form kernel values derivable from the reference source,
 *   - the invariants asserted in the reference's examples,
 *   - cross-checks against numpy/scipy LAPACK (tests/test_oracle_*.py).
 * Julia-stdlib behaviours that are not in /root/reference (mean, median, norm, dot, \,
 * cholesky, svd sign) are restated from the language's documented semantics and are
 * marked [stdlib] where used.
 *
 * Conventions: fp64 everywhere (the reference is effectively Float64-only,
 * RKHS.jl:4-11, partition.jl:135); points are packed point-major, X[d + D*i]
 * (the D x N column-major layout of array2matrix, utilities.jl:25-36); matrices are
 * column-major; all indices crossing this API are 0-based (the Julia side is 1-based).
 */
#ifndef PMK_ORACLE_H
#define PMK_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* kernel families (declarations.jl:18-45,65-67,75-111) */
enum {
    PMKO_SPLINE34 = 1,   /* Spline34KernelType(a)                 kernel.jl:299-313 */
    PMKO_SPLINE12 = 2,   /* Spline12KernelType(a)                 kernel.jl:316-330 */
    PMKO_SPLINE32 = 3,   /* Spline32KernelType(a)                 kernel.jl:333-347 */
    PMKO_GAUSSIAN = 4,   /* GaussianKernel1DType(eps_sq)          kernel.jl:350-357 */
    PMKO_RQ       = 5,   /* RationalQuadraticKernelType(a)        kernel.jl:360-366 */
    PMKO_TRQ      = 6,   /* TunableRationalQuadraticKernelType(a,w) kernel.jl:368-374 */
    PMKO_MODSQEXP = 7,   /* ModulatedSqExpKernelType(eps_sq, nu)  kernel.jl:376-391 */
    PMKO_BB10     = 10,  /* BrownianBridge10                      kernel.jl:156-158 */
    PMKO_BB20     = 11,  /* BrownianBridge20                      kernel.jl:218-225 */
    PMKO_BB1EPS   = 12,  /* BrownianBridge1eps(eps)               kernel.jl:168-174 */
    PMKO_BB2EPS   = 13   /* BrownianBridge2eps(eps)               kernel.jl:176-193 */
};
#define PMKO_FLAG_SEMIINF 1   /* BrownianBridgeSemiInfDomain wrapper, kernel.jl:256-263 */

typedef struct {
    int32_t family;
    int32_t flags;
    double  p[4];
} pmko_kernel;

/* ---- kernels ---- */
double pmko_profile(const pmko_kernel *th, double tau);                 /* evalkernel(tau, theta) */
double pmko_kernel_eval(const pmko_kernel *th, int D, const double *p, const double *q);
void   pmko_kernel_matrix(const pmko_kernel *th, int D, int64_t n, const double *X,
                          double *K, int64_t ldk);                      /* RKHS.jl:13-34 */
void   pmko_cross_kernel_matrix(const pmko_kernel *th, int D, int64_t n, const double *X,
                          int64_t m, const double *Z, double *K, int64_t ldk); /* RKHS.jl:95-110 */

/* ---- BSP ---- */
typedef struct pmko_bsp pmko_bsp;
/* sign_mode: +1 -> v = +z/|z| (LAPACK gesdd on the D x 1 parent, Julia's svd(adjoint));
 *            -1 -> v = -sign(z1) z/|z| (dense 1 x D path).  See SURVEY Appendix A.1. */
/* dot_mode: 0 -> dot(v, x) as separate multiplies and adds; 1 -> as a chain of fused multiply-adds (see dot_seq) */
pmko_bsp *pmko_bsp_build(int D, int64_t N, const double *X, int levels, int sign_mode, int dot_mode, int *status);
void      pmko_bsp_free(pmko_bsp *t);
int       pmko_bsp_levels(const pmko_bsp *t);
int       pmko_bsp_dim(const pmko_bsp *t);
int64_t   pmko_bsp_num_leaves(const pmko_bsp *t);
/* hyperplanes in pre-order (fetchhyperplanes, mixtureGP.jl:322-334): v is D x (P-1) */
void      pmko_bsp_hyperplanes(const pmko_bsp *t, double *v, double *c);
/* leaf l (0-based, left->right = AbstractTrees.Leaves order): offsets[P+1], inds[N] ascending */
void      pmko_bsp_leaves(const pmko_bsp *t, int64_t *offsets, int64_t *inds);
int64_t   pmko_bsp_findpartition(const pmko_bsp *t, const double *x);   /* partition.jl:248-262 */
/* organizetrainingsets (partition.jl:301-357).  Two calls: inds == NULL -> only counts.
 * counts[P]; then inds (sum counts) grouped by region via offsets[P+1]; per-point region
 * lists: list_offsets[N+1], lists[sum counts] (either may be NULL). */
void      pmko_bsp_assign(const pmko_bsp *t, int64_t N, const double *X, double eps,
                          int64_t *offsets, int64_t *inds,
                          int64_t *list_offsets, int64_t *lists);
/* findneighbourpartitions (mixtureGP.jl:339-405): returns number kept; region_inds[kept],
 * ts[P-1], zs[D x (P-1)], keep[P-1] (any of ts/zs/keep may be NULL). */
int64_t   pmko_bsp_neighbours(const pmko_bsp *t, const double *p, double radius, double delta,
                          int64_t home, int64_t *region_inds, double *ts, double *zs, uint8_t *keep);

/* ---- per-patch fit (one iteration of fitmixtureGP!, mixtureGP.jl:92-115) ----
 * K_out (n x n, K without noise = U_set entry) may be NULL.  c_lu = U\y by LU with partial
 * pivoting [stdlib]; L = cholesky(U).L (lower, col-major, ld n, strict upper zeroed).
 * c_chol (may be NULL) = L^-T L^-1 y, the value the GPU path computes.
 * returns 0, or k>0 when the leading minor k is not positive definite (PosDefException(k)),
 * or -1 for a singular LU. */
int pmko_fit_patch(const pmko_kernel *th, int D, int64_t n, const double *X, const double *y,
                   double sigma2, double *K_out, double *c_lu, double *L, double *c_chol);
int pmko_fit_patch_diag(const pmko_kernel *th, int D, int64_t n, const double *X, const double *y,
                        double sigma2, const double *diag, double *K_out, double *c_lu, double *L, double *c_chol);

/* fitRKHS! (RKHS.jl:182-217): c = (K + sigma2 I) \ y by LU */
int pmko_fit_rkhs(const pmko_kernel *th, int D, int64_t n, const double *X, const double *y,
                  double sigma2, double *c);
/* query! (RKHS.jl:220-247): mean only */
void pmko_query_rkhs(const pmko_kernel *th, int D, int64_t n, const double *X, const double *c,
                     int64_t nq, const double *Xq, double *Yq);

/* queryinner! (mixtureGP.jl:296-316); work: n doubles */
void pmko_queryinner(const pmko_kernel *th, int D, int64_t n, const double *X, const double *c,
                     const double *L, int64_t ldl, const double *xq, double min_v,
                     double *work, double *mu, double *var);
void pmko_queryinner_diag(const pmko_kernel *th, int D, int64_t n, const double *X, const double *c,
                          const double *L, int64_t ldl, const double *xq, double qdiag, double min_v,
                          double *kq, double *mu, double *var);

/* querymixtureGP! (mixtureGP.jl:159-294).  Model = per-region arrays.
 * Optional debug outputs (NULL to skip): home[Nq]; nb_offsets[Nq+1] + nb_regions/nb_t
 * (neighbour items in hyperplane order, capacity nb_cap; returns needed count). */
int64_t pmko_query_mixture(const pmko_bsp *t, const pmko_kernel *th, const pmko_kernel *weight_th,
                     const int64_t *n, const double *const *X, const double *const *c,
                     const double *const *L, int64_t Nq, const double *Xq,
                     double radius, double delta, double *Yq, double *Vq,
                     int64_t *home, int64_t *nb_offsets, int64_t *nb_regions, double *nb_t,
                     int64_t nb_cap, int nthreads);

/* low-level pieces exported for cross-checks */
void   pmko_mean_pairwise(int D, int64_t N, const double *X, double *mu);  /* [stdlib] mean */
double pmko_median(int64_t n, double *work);                               /* [stdlib] median! */
int    pmko_lu_solve(int64_t n, double *A, int64_t lda, double *b);         /* getrf+getrs, in place */
int    pmko_cholesky_lower(int64_t n, double *A, int64_t lda);              /* potrf 'L', in place */

#ifdef __cplusplus
}
#endif
#endif
