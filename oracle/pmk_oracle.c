/*
 * pmk_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See pmk_oracle.h.
 *
 * PARITY UNPINNED (no golden vectors exist in the reference; Julia is not available).
 *
 * Every function cites the reference file:line (relative to /root/reference) it restates.
 * Build with -ffp-contract=off: Julia never contracts a*b+c into an FMA, and the integer
 * outputs of the BSP (leaf ids, index lists, neighbour lists) depend on that.
 */
#include "pmk_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * Kernel profiles: evalkernel(tau, theta)
 * ---------------------------------------------------------------------------------------- */

/* src/RKHS/kernel.jl:299-313 (Spline34), :316-330 (Spline12), :333-347 (Spline32),
 * :350-357 (Gaussian), :360-366 (RQ), :368-374 (TRQ), :376-381 (ModulatedSqExp scalar form) */
double pmko_profile(const pmko_kernel *th, double tau)
{
    switch (th->family) {
    case PMKO_SPLINE34: {
        double r = tau * th->p[0];
        double tmp = 1.0 - r;
        if (tmp < 0.0) return 0.0;            /* sign(tmp) < 0 ; tmp == 0 passes and yields 0 */
        tmp = pow(tmp, 6.0);                  /* [stdlib] tmp^6 */
        return (((35.0 * (r * r) + 18.0 * r) + 3.0) * tmp) / 3.0;
    }
    case PMKO_SPLINE12: {
        double r = tau * th->p[0];
        double tmp = 1.0 - r;
        if (tmp < 0.0) return 0.0;
        tmp = tmp * tmp * tmp;                /* [stdlib] literal ^3 = x*x*x */
        return (3.0 * r + 1.0) * tmp;
    }
    case PMKO_SPLINE32: {
        double r = tau * th->p[0];
        double tmp = 1.0 - r;
        if (tmp < 0.0) return 0.0;
        tmp = pow(tmp, 4.0);                  /* [stdlib] tmp^4 */
        return (4.0 * r + 1.0) * tmp;
    }
    case PMKO_GAUSSIAN:
        return exp((-th->p[0]) * (tau * tau));
    case PMKO_RQ: {
        double s = sqrt(th->p[0] + tau * tau);
        double den = s * s * s;
        double sa = sqrt(th->p[0]);
        return (sa * sa * sa) / den;
    }
    case PMKO_TRQ: {
        double s = sqrt(th->p[0] + tau * tau);
        double den = s * s * s;
        double sa = sqrt(th->p[0]);
        return (th->p[1] * (sa * sa * sa)) / den;
    }
    case PMKO_MODSQEXP:
        return exp((-th->p[0]) * (tau * tau)) * cos(th->p[1] * tau);
    default:
        return NAN;
    }
}

/* Brownian-bridge scalar kernels on [0,1]: kernel.jl:156-158 (BB10), :218-225 (BB20),
 * :168-174 (BB1eps), :176-193 (BB2eps); semi-infinite pre-map :256-263 */
static double bb_scalar(const pmko_kernel *th, double x, double z)
{
    if (th->flags & PMKO_FLAG_SEMIINF) {
        x = x / (2.0 * (1.0 + x));
        z = z / (2.0 * (1.0 + z));
    }
    switch (th->family) {
    case PMKO_BB10:
        return fmin(x, z) - x * z;
    case PMKO_BB20: {
        const double m16 = -1.0 / 6.0;
        if (z < x)
            return ((m16 * z) * (1.0 - x)) * ((x * x + z * z) - 2.0 * x);
        return ((m16 * x) * (1.0 - z)) * ((x * x + z * z) - 2.0 * z);
    }
    case PMKO_BB1EPS: {
        double e = th->p[0];
        double den = e * sinh(e);
        double num = sinh(e * fmin(x, z)) * sinh(e * (1.0 - fmax(x, z)));
        return num / den;
    }
    case PMKO_BB2EPS: {
        double e = th->p[0];
        double s = x + z;
        double mn = fmin(x, z), mx = fmax(x, z), ad = fabs(x - z);
        double num = exp((-e) * s);
        double em1 = exp(2.0 * e) - 1.0;
        double den = (4.0 * (e * e * e)) * (em1 * em1);
        double mult = num / den;
        double t1 = exp(2.0 * e) * ((2.0 * e - e * s) - 1.0);
        double t2 = exp(4.0 * e) * (e * s + 1.0);
        double t3 = exp((2.0 * e) * ((1.0 + x) + z)) * ((2.0 * e - e * s) + 1.0);
        double t4 = exp((2.0 * e) * s) * (e * s - 1.0);
        double t5 = exp((2.0 * e) * (2.0 + mn)) * ((-e) * ad - 1.0);
        double t6 = exp((2.0 * e) * mx) * ((-e) * ad + 1.0);
        double t7 = exp((2.0 * e) * (1.0 + mn)) * ((1.0 - 2.0 * e) + e * ad);
        double t8 = exp((2.0 * e) * (1.0 + mx)) * ((1.0 + 2.0 * e) - e * ad);
        return mult * (((((((t1 + t2) + t3) + t4) + t5) + t6) + t7) + t8);
    }
    default:
        return NAN;
    }
}

/* [stdlib] LinearAlgebra.norm of a short vector (generic_norm2): max-abs guard, then the
 * sequential sum of squares and one sqrt; the scaled branch only on under/overflow.
 * Called on x1-x2 by kernel.jl:283. */
static double norm2_diff(int D, const double *p, const double *q)
{
    double maxabs = 0.0;
    for (int d = 0; d < D; ++d) {
        double a = fabs(p[d] - q[d]);
        if (a > maxabs || isnan(a)) maxabs = a;
    }
    if (maxabs == 0.0 || isinf(maxabs)) return maxabs;
    double mm = maxabs * maxabs;
    if (isfinite((double)D * mm) && mm != 0.0) {
        double s = 0.0;
        for (int d = 0; d < D; ++d) {
            double r = p[d] - q[d];
            s = (d == 0) ? r * r : s + r * r;
        }
        return sqrt(s);
    }
    double s = 0.0;
    for (int d = 0; d < D; ++d) {
        double r = (p[d] - q[d]) / maxabs;
        s = (d == 0) ? r * r : s + r * r;
    }
    return maxabs * sqrt(s);
}

/* evalkernel(p, q, theta): stationary dispatch kernel.jl:277-287 (tau = norm(x1-x2)),
 * Brownian-bridge tensor product kernel.jl:196-206 */
double pmko_kernel_eval(const pmko_kernel *th, int D, const double *p, const double *q)
{
    if (th->family >= PMKO_BB10) {
        double out = bb_scalar(th, p[0], q[0]);
        for (int d = 1; d < D; ++d) out = out * bb_scalar(th, p[d], q[d]);
        return out;
    }
    if (th->family == PMKO_MODSQEXP && D > 1) return NAN;   /* kernel.jl:383-391 errors for D>1 */
    return pmko_profile(th, norm2_diff(D, p, q));
}

/* constructkernelmatrix! src/RKHS/RKHS.jl:13-34: lower triangle incl. diagonal with the ROW
 * point as first argument, then mirror -> exactly symmetric */
void pmko_kernel_matrix(const pmko_kernel *th, int D, int64_t n, const double *X,
                        double *K, int64_t ldk)
{
    for (int64_t j = 0; j < n; ++j)
        for (int64_t i = j; i < n; ++i)
            K[i + j * ldk] = pmko_kernel_eval(th, D, X + i * D, X + j * D);
    for (int64_t j = 1; j < n; ++j)
        for (int64_t i = 0; i < j; ++i)
            K[i + j * ldk] = K[j + i * ldk];
}

/* constructkernelmatrix(X, Z, theta) src/RKHS/RKHS.jl:95-110 */
void pmko_cross_kernel_matrix(const pmko_kernel *th, int D, int64_t n, const double *X,
                              int64_t m, const double *Z, double *K, int64_t ldk)
{
    for (int64_t j = 0; j < m; ++j)
        for (int64_t i = 0; i < n; ++i)
            K[i + j * ldk] = pmko_kernel_eval(th, D, X + i * D, Z + j * D);
}

/* ------------------------------------------------------------------------------------------
 * [stdlib] Statistics.mean / median restated
 * ---------------------------------------------------------------------------------------- */

/* sum of the vectors X[idx[first..last]] as Base.mapreduce_impl does it (pairwise, block 1024) */
static void pairwise_sum(int D, const double *X, const int64_t *idx, int64_t first, int64_t last,
                         double *out)
{
    if (first == last) {
        const double *x = X + (idx ? idx[first] : first) * D;
        for (int d = 0; d < D; ++d) out[d] = x[d];
        return;
    }
    if (last - first < 1024) {
        const double *a = X + (idx ? idx[first] : first) * D;
        const double *b = X + (idx ? idx[first + 1] : first + 1) * D;
        for (int d = 0; d < D; ++d) out[d] = a[d] + b[d];
        for (int64_t i = first + 2; i <= last; ++i) {
            const double *x = X + (idx ? idx[i] : i) * D;
            for (int d = 0; d < D; ++d) out[d] = out[d] + x[d];
        }
        return;
    }
    int64_t mid = first + ((last - first) >> 1);
    double tmp[16];
    pairwise_sum(D, X, idx, first, mid, out);
    pairwise_sum(D, X, idx, mid + 1, last, tmp);
    for (int d = 0; d < D; ++d) out[d] = out[d] + tmp[d];
}

static void mean_subset(int D, const double *X, const int64_t *idx, int64_t n, double *mu)
{
    pairwise_sum(D, X, idx, 0, n - 1, mu);
    for (int d = 0; d < D; ++d) mu[d] = mu[d] / (double)n;
}

void pmko_mean_pairwise(int D, int64_t N, const double *X, double *mu)
{
    mean_subset(D, X, NULL, N, mu);
}

static int cmp_double(const void *a, const void *b)
{
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

/* [stdlib] median!: odd n -> middle order statistic; even n -> a/2 + b/2 (Statistics.middle) */
double pmko_median(int64_t n, double *v)
{
    qsort(v, (size_t)n, sizeof(double), cmp_double);
    if (n & 1) return v[n / 2];
    return v[n / 2 - 1] / 2.0 + v[n / 2] / 2.0;
}

/* ------------------------------------------------------------------------------------------
 * BSP tree: src/patchwork/partition.jl
 * Complete binary tree of depth levels-1 kept in heap order (root 0, children 2i+1, 2i+2).
 * ---------------------------------------------------------------------------------------- */
struct pmko_bsp {
    int D, levels;
    int dot_mode;       /* see dot_seq */
    int64_t P;          /* leaves = 2^(levels-1) */
    int64_t N;
    double *v;          /* (P-1) x D, heap order */
    double *c;          /* P-1 */
    int64_t *leaf_off;  /* P+1 */
    int64_t *leaf_inds; /* N, grouped by leaf, ascending inside a leaf */
    int64_t *pre;       /* pre-order rank -> heap index (P-1) */
};

static double dot_seq(int D, const double *a, const double *b, int mode)
{
    /* [stdlib] dot on a short Vector{Float64} reaches BLAS ddot: a sequential loop whose multiply-add is either left
     * as two operations (mode 0, default) or contracted into fused multiply-adds (mode 1) by the BLAS build.  Which one
     * a Julia install runs is not decidable from the reference's text: both are restated, the product has the same
     * switch, and the tests hold the two implementations to each other in both modes. */
    double s = a[0] * b[0];
    if (mode) { for (int d = 1; d < D; ++d) s = fma(a[d], b[d], s); }
    else { for (int d = 1; d < D; ++d) s = s + a[d] * b[d]; }
    return s;
}

/* gethyperplane partition.jl:86-100 + splitpoints :64-83.  idx: the node's points in original
 * order.  Writes v, c, and the left mask. */
static int gethyperplane(int D, const double *X, const int64_t *idx, int64_t n, int sign_mode, int dot_mode,
                         double *v, double *c, uint8_t *left)
{
    double mu[16], z[16];
    if (n <= 0) return -1;
    mean_subset(D, X, idx, n, mu);                                   /* :89 */
    const double *x1 = X + idx[0] * D;                               /* :90  size(X,2)==1 quirk */
    double s = 0.0;
    for (int d = 0; d < D; ++d) { z[d] = x1[d] - mu[d]; s = (d == 0) ? z[d] * z[d] : s + z[d] * z[d]; }
    double nz = sqrt(s);
    if (nz == 0.0) {                                                 /* svd of a zero row: V = I */
        for (int d = 0; d < D; ++d) v[d] = (d == 0) ? 1.0 : 0.0;
    } else {
        double sg = 1.0;                                             /* [stdlib] :92-94 */
        if (sign_mode < 0) sg = (z[0] > 0.0) ? -1.0 : 1.0;
        for (int d = 0; d < D; ++d) v[d] = sg * (z[d] / nz);
    }
    double *ev = (double *)malloc(sizeof(double) * (size_t)n * 2);
    if (!ev) return -2;
    for (int64_t i = 0; i < n; ++i) ev[i] = dot_seq(D, v, X + idx[i] * D, dot_mode);   /* :69 */
    memcpy(ev + n, ev, sizeof(double) * (size_t)n);
    *c = pmko_median(n, ev + n);                                     /* :70 */
    for (int64_t i = 0; i < n; ++i) left[i] = ev[i] < *c;             /* :72-80 strict */
    free(ev);
    return 0;
}

/* createchildren partition.jl:166-217, recursion flattened onto heap indices */
static int build_rec(pmko_bsp *t, const double *X, int sign_mode, int64_t node, int depth,
                     int64_t *idx, int64_t n, int64_t **leaf_lists, int64_t *leaf_counts)
{
    if (depth == t->levels - 1) {                                   /* leaf (:192-201) */
        int64_t leaf = node - (t->P - 1);
        leaf_lists[leaf] = idx;
        leaf_counts[leaf] = n;
        return 0;
    }
    uint8_t *left = (uint8_t *)malloc((size_t)(n > 0 ? n : 1));
    if (!left) return -2;
    int st = gethyperplane(t->D, X, idx, n, sign_mode, t->dot_mode, t->v + node * t->D, t->c + node, left);
    if (st) { free(left); free(idx); return st; }
    int64_t nl = 0;
    for (int64_t i = 0; i < n; ++i) nl += left[i];
    int64_t *li = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nl > 0 ? nl : 1));
    int64_t *ri = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n - nl > 0 ? n - nl : 1));
    int64_t a = 0, b = 0;
    for (int64_t i = 0; i < n; ++i) {                                /* mask indexing keeps order */
        if (left[i]) li[a++] = idx[i]; else ri[b++] = idx[i];
    }
    free(left);
    free(idx);
    st = build_rec(t, X, sign_mode, 2 * node + 1, depth + 1, li, nl, leaf_lists, leaf_counts);
    if (st) { free(ri); return st; }
    return build_rec(t, X, sign_mode, 2 * node + 2, depth + 1, ri, n - nl, leaf_lists, leaf_counts);
}

static void preorder(const pmko_bsp *t, int64_t node, int depth, int64_t *k)
{
    if (depth == t->levels - 1) return;
    t->pre[(*k)++] = node;
    preorder(t, 2 * node + 1, depth + 1, k);
    preorder(t, 2 * node + 2, depth + 1, k);
}

/* setuppartition partition.jl:106-129 + labelleafnodes :131-159 */
pmko_bsp *pmko_bsp_build(int D, int64_t N, const double *X, int levels, int sign_mode, int dot_mode, int *status)
{
    int st = 0;
    if (D < 1 || D > 16 || levels < 2 || levels > 40 || N < 1) { if (status) *status = -1; return NULL; }
    pmko_bsp *t = (pmko_bsp *)calloc(1, sizeof(*t));
    t->D = D; t->levels = levels; t->N = N; t->dot_mode = dot_mode != 0;
    t->P = (int64_t)1 << (levels - 1);
    t->v = (double *)calloc((size_t)((t->P - 1) * D), sizeof(double));
    t->c = (double *)calloc((size_t)(t->P - 1), sizeof(double));
    t->leaf_off = (int64_t *)calloc((size_t)(t->P + 1), sizeof(int64_t));
    t->leaf_inds = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    t->pre = (int64_t *)malloc(sizeof(int64_t) * (size_t)(t->P - 1));
    int64_t **lists = (int64_t **)calloc((size_t)t->P, sizeof(int64_t *));
    int64_t *counts = (int64_t *)calloc((size_t)t->P, sizeof(int64_t));
    int64_t *idx = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    for (int64_t i = 0; i < N; ++i) idx[i] = i;
    st = build_rec(t, X, sign_mode, 0, 0, idx, N, lists, counts);
    if (!st) {
        for (int64_t l = 0; l < t->P; ++l) {
            t->leaf_off[l + 1] = t->leaf_off[l] + counts[l];
            memcpy(t->leaf_inds + t->leaf_off[l], lists[l], sizeof(int64_t) * (size_t)counts[l]);
        }
        int64_t k = 0;
        preorder(t, 0, 0, &k);
    }
    for (int64_t l = 0; l < t->P; ++l) free(lists[l]);
    free(lists); free(counts);
    if (status) *status = st;
    if (st) { pmko_bsp_free(t); return NULL; }
    return t;
}

void pmko_bsp_free(pmko_bsp *t)
{
    if (!t) return;
    free(t->v); free(t->c); free(t->leaf_off); free(t->leaf_inds); free(t->pre); free(t);
}

int     pmko_bsp_levels(const pmko_bsp *t) { return t->levels; }
int     pmko_bsp_dim(const pmko_bsp *t) { return t->D; }
int64_t pmko_bsp_num_leaves(const pmko_bsp *t) { return t->P; }

/* fetchhyperplanes src/RKHS/mixtureGP.jl:322-334: PreOrderDFS over internal nodes */
void pmko_bsp_hyperplanes(const pmko_bsp *t, double *v, double *c)
{
    for (int64_t k = 0; k < t->P - 1; ++k) {
        int64_t h = t->pre[k];
        for (int d = 0; d < t->D; ++d) v[d + k * t->D] = t->v[d + h * t->D];
        c[k] = t->c[h];
    }
}

void pmko_bsp_leaves(const pmko_bsp *t, int64_t *offsets, int64_t *inds)
{
    memcpy(offsets, t->leaf_off, sizeof(int64_t) * (size_t)(t->P + 1));
    if (inds) memcpy(inds, t->leaf_inds, sizeof(int64_t) * (size_t)t->N);
}

/* findpartition partition.jl:248-262 */
int64_t pmko_bsp_findpartition(const pmko_bsp *t, const double *x)
{
    int64_t node = 0;
    for (int l = 1; l <= t->levels - 1; ++l) {
        if (dot_seq(t->D, t->v + node * t->D, x, t->dot_mode) < t->c[node]) node = 2 * node + 1;
        else node = 2 * node + 2;
    }
    return node - (t->P - 1);
}

/* find-eps-partitions partition.jl:269-298 (left first, then right) */
static void find_eps(const pmko_bsp *t, const double *x, int64_t node, int depth, double eps,
                     int64_t *list, int64_t *cnt)
{
    if (depth == t->levels - 1) { list[(*cnt)++] = node - (t->P - 1); return; }
    double e = dot_seq(t->D, t->v + node * t->D, x, t->dot_mode);
    if (e < t->c[node] + eps) find_eps(t, x, 2 * node + 1, depth + 1, eps, list, cnt);
    if (e > t->c[node] - eps) find_eps(t, x, 2 * node + 2, depth + 1, eps, list, cnt);
}

/* organizetrainingsets partition.jl:301-357 */
void pmko_bsp_assign(const pmko_bsp *t, int64_t N, const double *X, double eps,
                     int64_t *offsets, int64_t *inds, int64_t *list_offsets, int64_t *lists)
{
    int64_t *list = (int64_t *)malloc(sizeof(int64_t) * (size_t)t->P);
    int64_t *fill = (int64_t *)calloc((size_t)t->P, sizeof(int64_t));
    for (int64_t r = 0; r <= t->P; ++r) offsets[r] = 0;
    for (int64_t n = 0; n < N; ++n) {
        int64_t cnt = 0;
        find_eps(t, X + n * t->D, 0, 0, eps, list, &cnt);
        for (int64_t m = 0; m < cnt; ++m) offsets[list[m] + 1]++;
    }
    for (int64_t r = 0; r < t->P; ++r) offsets[r + 1] += offsets[r];
    if (list_offsets) list_offsets[0] = 0;
    if (inds || list_offsets || lists) {
        int64_t tot = 0;
        for (int64_t n = 0; n < N; ++n) {
            int64_t cnt = 0;
            find_eps(t, X + n * t->D, 0, 0, eps, list, &cnt);
            for (int64_t m = 0; m < cnt; ++m) {
                int64_t r = list[m];
                if (inds) inds[offsets[r] + fill[r]] = n;
                fill[r]++;
                if (lists) lists[tot + m] = r;
            }
            tot += cnt;
            if (list_offsets) list_offsets[n + 1] = tot;
        }
    }
    free(list); free(fill);
}

/* findneighbourpartitions src/RKHS/mixtureGP.jl:339-405 */
int64_t pmko_bsp_neighbours(const pmko_bsp *t, const double *p, double radius, double delta,
                            int64_t home, int64_t *region_inds, double *ts, double *zs, uint8_t *keep)
{
    int D = t->D;
    int64_t j = 0;
    double z[16], z1[16], z2[16];
    for (int64_t i = 0; i < t->P - 1; ++i) {
        const double *u = t->v + t->pre[i] * D;
        double c = t->c[t->pre[i]];
        double tt = -dot_seq(D, u, p, t->dot_mode) + c;                /* :361 */
        for (int d = 0; d < D; ++d) z[d] = p[d] + tt * u[d];          /* :362 */
        if (ts) ts[i] = tt;
        if (zs) for (int d = 0; d < D; ++d) zs[d + i * D] = z[d];
        if (keep) keep[i] = 0;
        if (norm2_diff(D, z, p) < radius) {                           /* :367 */
            double tp = tt + delta, tm = tt - delta;
            for (int d = 0; d < D; ++d) { z1[d] = p[d] + tp * u[d]; z2[d] = p[d] + tm * u[d]; }
            int64_t r1 = pmko_bsp_findpartition(t, z1);               /* :374-375 */
            int64_t r2 = pmko_bsp_findpartition(t, z2);
            if ((r2 == home) != (r1 == home)) {                       /* xor :388 */
                if (keep) keep[i] = 1;
                region_inds[j++] = (r1 == home) ? r2 : r1;            /* :392-396 */
            }
        }
    }
    return j;
}

/* ------------------------------------------------------------------------------------------
 * Dense solves.  [stdlib] U\y on a square dense matrix = LU with partial pivoting
 * (mixtureGP.jl:106, RKHS.jl:214); cholesky(U) (mixtureGP.jl:109); L\kq = trsv (:311).
 * Unblocked LAPACK-style loops (getf2 / potf2 / trsv); OpenBLAS' blocking is unpinned.
 * ---------------------------------------------------------------------------------------- */
int pmko_lu_solve(int64_t n, double *A, int64_t lda, double *b)
{
    int64_t *piv = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    int info = 0;
    for (int64_t j = 0; j < n; ++j) {
        int64_t p = j; double mx = fabs(A[j + j * lda]);
        for (int64_t i = j + 1; i < n; ++i) { double a = fabs(A[i + j * lda]); if (a > mx) { mx = a; p = i; } }
        piv[j] = p;
        if (mx == 0.0) { if (!info) info = -1; continue; }
        if (p != j)
            for (int64_t k = 0; k < n; ++k) { double tmp = A[j + k * lda]; A[j + k * lda] = A[p + k * lda]; A[p + k * lda] = tmp; }
        double inv = 1.0 / A[j + j * lda];
        for (int64_t i = j + 1; i < n; ++i) A[i + j * lda] *= inv;
        for (int64_t k = j + 1; k < n; ++k) {
            double ujk = A[j + k * lda];
            if (ujk != 0.0) {
                double *col = A + k * lda; const double *l = A + j * lda;
                for (int64_t i = j + 1; i < n; ++i) col[i] -= l[i] * ujk;
            }
        }
    }
    if (!info) {
        for (int64_t j = 0; j < n; ++j) if (piv[j] != j) { double tmp = b[j]; b[j] = b[piv[j]]; b[piv[j]] = tmp; }
        for (int64_t j = 0; j < n; ++j) {                 /* unit lower, column oriented */
            double bj = b[j];
            if (bj != 0.0) for (int64_t i = j + 1; i < n; ++i) b[i] -= bj * A[i + j * lda];
        }
        for (int64_t j = n - 1; j >= 0; --j) {            /* upper */
            b[j] /= A[j + j * lda];
            double bj = b[j];
            for (int64_t i = 0; i < j; ++i) b[i] -= bj * A[i + j * lda];
        }
    }
    free(piv);
    return info;
}

int pmko_cholesky_lower(int64_t n, double *A, int64_t lda)
{
    for (int64_t j = 0; j < n; ++j) {
        double *cj = A + j * lda;
        for (int64_t k = 0; k < j; ++k) {
            const double *ck = A + k * lda;
            double ljk = ck[j];
            if (ljk != 0.0) for (int64_t i = j; i < n; ++i) cj[i] -= ck[i] * ljk;
        }
        double d = cj[j];
        if (!(d > 0.0)) return (int)(j + 1);              /* PosDefException(j+1) */
        d = sqrt(d);
        cj[j] = d;
        for (int64_t i = j + 1; i < n; ++i) cj[i] /= d;
    }
    return 0;
}

/* x <- L^-1 x, reference-BLAS dtrsv('L','N','N') loop order */
static void trsv_lower(int64_t n, const double *L, int64_t ldl, double *x)
{
    for (int64_t j = 0; j < n; ++j) {
        if (x[j] != 0.0) {
            x[j] /= L[j + j * ldl];
            double xj = x[j];
            const double *col = L + j * ldl;
            for (int64_t i = j + 1; i < n; ++i) x[i] -= xj * col[i];
        }
    }
}

/* x <- L^-T x, dtrsv('L','T','N') */
static void trsv_lower_t(int64_t n, const double *L, int64_t ldl, double *x)
{
    for (int64_t j = n - 1; j >= 0; --j) {
        double tmp = x[j];
        const double *col = L + j * ldl;
        for (int64_t i = n - 1; i > j; --i) tmp -= col[i] * x[i];
        x[j] = tmp / col[j];
    }
}

/* one iteration of fitmixtureGP! src/RKHS/mixtureGP.jl:92-115.
 * diag (may be NULL): the point-dependent diagonal term of the reference's DPP kernels -- evalkernel(p, p, theta) =
 * 1 + g(p) for AdaptiveKernelDPPType / AdaptiveKernelMultiWarpDPPType (src/RKHS/kernel.jl:70-75, 102-110), whose
 * off-diagonal part is a stationary kernel on positions + warp values (X is then the augmented point set): part of K,
 * added before the noise. */
int pmko_fit_patch_diag(const pmko_kernel *th, int D, int64_t n, const double *X, const double *y,
                        double sigma2, const double *diag, double *K_out, double *c_lu, double *L, double *c_chol)
{
    int info = 0;
    pmko_kernel_matrix(th, D, n, X, L, n);                             /* :98 */
    if (diag)
        for (int64_t i = 0; i < n; ++i) L[i + i * n] += diag[i];       /* kernel.jl:74 / :108 */
    if (K_out) memcpy(K_out, L, sizeof(double) * (size_t)(n * n));    /* :99 U_set */
    for (int64_t i = 0; i < n; ++i) L[i + i * n] += sigma2;            /* :102-104 */
    if (c_lu) {                                                        /* :106 c = U\y */
        double *A = (double *)malloc(sizeof(double) * (size_t)(n * n));
        memcpy(A, L, sizeof(double) * (size_t)(n * n));
        memcpy(c_lu, y, sizeof(double) * (size_t)n);
        int st = pmko_lu_solve(n, A, n, c_lu);
        free(A);
        if (st) info = st;
    }
    int st = pmko_cholesky_lower(n, L, n);                             /* :109 */
    if (st) return st;
    for (int64_t j = 1; j < n; ++j)                                    /* .L : strict upper = 0 */
        for (int64_t i = 0; i < j; ++i) L[i + j * n] = 0.0;
    if (c_chol) {
        memcpy(c_chol, y, sizeof(double) * (size_t)n);
        trsv_lower(n, L, n, c_chol);
        trsv_lower_t(n, L, n, c_chol);
    }
    return info;
}

int pmko_fit_patch(const pmko_kernel *th, int D, int64_t n, const double *X, const double *y,
                   double sigma2, double *K_out, double *c_lu, double *L, double *c_chol)
{
    return pmko_fit_patch_diag(th, D, n, X, y, sigma2, NULL, K_out, c_lu, L, c_chol);
}

/* fitRKHS! src/RKHS/RKHS.jl:195-217 */
int pmko_fit_rkhs(const pmko_kernel *th, int D, int64_t n, const double *X, const double *y,
                  double sigma2, double *c)
{
    double *U = (double *)malloc(sizeof(double) * (size_t)(n * n));
    pmko_kernel_matrix(th, D, n, X, U, n);
    for (int64_t i = 0; i < n; ++i) U[i + i * n] += sigma2;
    memcpy(c, y, sizeof(double) * (size_t)n);
    int st = pmko_lu_solve(n, U, n, c);
    free(U);
    return st;
}

/* query! src/RKHS/RKHS.jl:220-247 */
void pmko_query_rkhs(const pmko_kernel *th, int D, int64_t n, const double *X, const double *c,
                     int64_t nq, const double *Xq, double *Yq)
{
    for (int64_t iq = 0; iq < nq; ++iq) {
        double s = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            double k = pmko_kernel_eval(th, D, Xq + iq * D, X + i * D);
            s = (i == 0) ? k * c[i] : s + k * c[i];
        }
        Yq[iq] = s;
    }
}

/* queryinner! src/RKHS/mixtureGP.jl:296-316; qdiag = the DPP kernels' diagonal term at the query point (0 otherwise) */
void pmko_queryinner_diag(const pmko_kernel *th, int D, int64_t n, const double *X, const double *c,
                          const double *L, int64_t ldl, const double *xq, double qdiag, double min_v,
                          double *kq, double *mu, double *var);

void pmko_queryinner(const pmko_kernel *th, int D, int64_t n, const double *X, const double *c,
                     const double *L, int64_t ldl, const double *xq, double min_v,
                     double *kq, double *mu, double *var)
{
    pmko_queryinner_diag(th, D, n, X, c, L, ldl, xq, 0.0, min_v, kq, mu, var);
}

void pmko_queryinner_diag(const pmko_kernel *th, int D, int64_t n, const double *X, const double *c,
                          const double *L, int64_t ldl, const double *xq, double qdiag, double min_v,
                          double *kq, double *mu, double *var)
{
    for (int64_t i = 0; i < n; ++i) kq[i] = pmko_kernel_eval(th, D, xq, X + i * D);   /* :302-305 */
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) s = (i == 0) ? kq[i] * c[i] : s + kq[i] * c[i];  /* :308 */
    *mu = s;
    trsv_lower(n, L, ldl, kq);                                                        /* :311 */
    double vv = 0.0;
    for (int64_t i = 0; i < n; ++i) vv = (i == 0) ? kq[i] * kq[i] : vv + kq[i] * kq[i];
    double kself = pmko_kernel_eval(th, D, xq, xq);
    if (qdiag != 0.0) kself = kself + qdiag;                                           /* kernel.jl:74 / :108 */
    double vq = kself - vv;                                                            /* :312 */
    *var = (vq < min_v) ? min_v : vq;
}

/* querymixtureGP! src/RKHS/mixtureGP.jl:159-294 */
int64_t pmko_query_mixture(const pmko_bsp *t, const pmko_kernel *th, const pmko_kernel *wth,
                           const int64_t *n, const double *const *X, const double *const *c,
                           const double *const *L, int64_t Nq, const double *Xq,
                           double radius, double delta, double *Yq, double *Vq,
                           int64_t *home, int64_t *nb_offsets, int64_t *nb_regions, double *nb_t,
                           int64_t nb_cap, int nthreads)
{
    int D = t->D;
    int64_t P = t->P;
    int64_t nmax = 0;
    for (int64_t r = 0; r < P; ++r) if (n[r] > nmax) nmax = n[r];
    int64_t *cnt = (int64_t *)calloc((size_t)(Nq + 1), sizeof(int64_t));
    /* pass 1 (cheap): neighbour counts -> offsets, so the debug arrays can be filled in parallel */
    (void)nthreads;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
#endif
    {
        int64_t *reg = (int64_t *)malloc(sizeof(int64_t) * (size_t)(P > 1 ? P - 1 : 1));
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int64_t j = 0; j < Nq; ++j) {
            int64_t h = pmko_bsp_findpartition(t, Xq + j * D);
            cnt[j + 1] = pmko_bsp_neighbours(t, Xq + j * D, radius, delta, h, reg, NULL, NULL, NULL);
        }
        free(reg);
    }
    for (int64_t j = 0; j < Nq; ++j) cnt[j + 1] += cnt[j];
    int64_t total = cnt[Nq];
    if (nb_offsets) memcpy(nb_offsets, cnt, sizeof(int64_t) * (size_t)(Nq + 1));
    int store_nb = (nb_regions != NULL) && (total <= nb_cap);

#ifdef _OPENMP
#pragma omp parallel
#endif
    {
        int64_t *reg = (int64_t *)malloc(sizeof(int64_t) * (size_t)(P > 1 ? P - 1 : 1));
        double *ts = (double *)malloc(sizeof(double) * (size_t)(P > 1 ? P - 1 : 1));
        uint8_t *keep = (uint8_t *)malloc((size_t)(P > 1 ? P - 1 : 1));
        double *w = (double *)malloc(sizeof(double) * (size_t)P * 3);
        double *u = w + P, *v = w + 2 * P;
        double *kq = (double *)malloc(sizeof(double) * (size_t)(nmax > 0 ? nmax : 1));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
        for (int64_t j = 0; j < Nq; ++j) {
            const double *xq = Xq + j * D;
            int64_t h = pmko_bsp_findpartition(t, xq);                                 /* :208 */
            int64_t nr = pmko_bsp_neighbours(t, xq, radius, delta, h, reg, ts, NULL, keep); /* :211 */
            int64_t k = 0;
            for (int64_t i = 0; i < P - 1; ++i) {
                if (!keep[i]) continue;                                                /* t_kept :213 */
                int64_t r = reg[k];
                w[k] = pmko_profile(wth, fabs(ts[i]));                                /* :231 */
                pmko_queryinner(th, D, n[r], X[r], c[r], L[r], n[r], xq, 1e-12, kq, &u[k], &v[k]);
                if (store_nb) { nb_regions[cnt[j] + k] = r; if (nb_t) nb_t[cnt[j] + k] = ts[i]; }
                ++k;
            }
            (void)nr;
            w[k] = 1.0;                                                                /* :237 */
            pmko_queryinner(th, D, n[h], X[h], c[h], L[h], n[h], xq, 1e-12, kq, &u[k], &v[k]);
            ++k;
            double sw = w[0];                                                          /* :263 sum(w) */
            for (int64_t i = 1; i < k; ++i) sw = sw + w[i];
            for (int64_t i = 0; i < k; ++i) w[i] = w[i] / sw;
            double yq = w[0] * u[0], vq = w[0] * (v[0] * w[0]);                        /* :269,:272 */
            for (int64_t i = 1; i < k; ++i) { yq = yq + w[i] * u[i]; vq = vq + w[i] * (v[i] * w[i]); }
            Yq[j] = yq; Vq[j] = vq;
            if (home) home[j] = h;
        }
        free(reg); free(ts); free(keep); free(w); free(kq);
    }
    free(cnt);
    return total;
}
