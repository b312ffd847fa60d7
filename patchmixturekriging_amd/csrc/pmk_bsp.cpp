// Host side of the BSP partitioner: exact (bit-reproducible) restatement of
// src/patchwork/partition.jl and of the partition searches in src/RKHS/mixtureGP.jl of the
// reference.  Level-by-level array formulation (no pointer tree): the points of every node are a
// contiguous segment of one permutation array that is stably partitioned in place, which keeps the
// reference's "mask indexing preserves the original order" property (partition.jl:177-186).
//
// Compiled with -ffp-contract=off: the leaf ids, index lists and neighbour lists are integer
// outputs of floating-point comparisons and must not depend on FMA contraction.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "pmk_internal.h"

namespace pmk {

// [Julia stdlib] dot(::Vector{Float64}, ::Vector{Float64}) reaches BLAS ddot; for the 1..4-element vectors of this
// path that is a short sequential loop whose multiply-add may or may not have been contracted by the BLAS build:
// mode 0 = separate multiply and add (the default), mode 1 = a chain of fused multiply-adds.  Both are exact
// restatements of *a* ddot; which one a given Julia install runs cannot be decided without running it.
static inline double dot_seq(int D, const double *a, const double *b, int mode)
{
    double s = a[0] * b[0];
    if (mode) { for (int d = 1; d < D; ++d) s = std::fma(a[d], b[d], s); }
    else { for (int d = 1; d < D; ++d) s = s + a[d] * b[d]; }
    return s;
}

// [Julia stdlib] sum of a Vector{Vector}: Base.mapreduce_impl, pairwise with 1024-element leaves
static void pairwise_sum(int D, const double *X, const int64_t *idx, int64_t first, int64_t last, double *out)
{
    if (first == last) {
        const double *x = X + idx[first] * D;
        for (int d = 0; d < D; ++d) out[d] = x[d];
        return;
    }
    if (last - first < 1024) {
        const double *a = X + idx[first] * D, *b = X + idx[first + 1] * D;
        for (int d = 0; d < D; ++d) out[d] = a[d] + b[d];
        for (int64_t i = first + 2; i <= last; ++i) {
            const double *x = X + idx[i] * D;
            for (int d = 0; d < D; ++d) out[d] = out[d] + x[d];
        }
        return;
    }
    const int64_t mid = first + ((last - first) >> 1);
    double tmp[MAX_D];
    pairwise_sum(D, X, idx, first, mid, out);
    pairwise_sum(D, X, idx, mid + 1, last, tmp);
    for (int d = 0; d < D; ++d) out[d] = out[d] + tmp[d];
}

// gethyperplane (partition.jl:89-94) from the node's coordinate sum: mu = sum / n, z = x1 - mu, v = +-z/|z|
void bsp_direction(int D, const double *sum, int64_t n, const double *x1, int sign_mode, double *v)
{
    double z[MAX_D];
    double s = 0.0;
    for (int d = 0; d < D; ++d) {
        const double mu = sum[d] / (double)n;
        z[d] = x1[d] - mu;
        s = (d == 0) ? z[d] * z[d] : s + z[d] * z[d];
    }
    const double nz = std::sqrt(s);
    if (nz == 0.0) {
        for (int d = 0; d < D; ++d) v[d] = (d == 0) ? 1.0 : 0.0;
    } else {
        double sg = 1.0;
        if (sign_mode < 0) sg = (z[0] > 0.0) ? -1.0 : 1.0;
        for (int d = 0; d < D; ++d) v[d] = sg * (z[d] / nz);
    }
}

// the same pairwise recursion split in two for the device build: the sequential leaf blocks [first, last] in
// visiting order (summed on the GPU, one thread per block) ...
void bsp_pairwise_blocks(int64_t first, int64_t last, int64_t base, std::vector<int64_t> &blk_first,
                         std::vector<int64_t> &blk_last)
{
    if (last - first < 1024) {
        blk_first.push_back(base + first);
        blk_last.push_back(base + last);
        return;
    }
    const int64_t mid = first + ((last - first) >> 1);
    bsp_pairwise_blocks(first, mid, base, blk_first, blk_last);
    bsp_pairwise_blocks(mid + 1, last, base, blk_first, blk_last);
}

// ... and the combination of their partial sums in the recursion's order (host)
void bsp_pairwise_combine(int D, int64_t first, int64_t last, const double *partials, int64_t &cursor, double *out)
{
    if (last - first < 1024) {
        const double *p = partials + (cursor++) * D;
        for (int d = 0; d < D; ++d) out[d] = p[d];
        return;
    }
    const int64_t mid = first + ((last - first) >> 1);
    double tmp[MAX_D];
    bsp_pairwise_combine(D, first, mid, partials, cursor, out);
    bsp_pairwise_combine(D, mid + 1, last, partials, cursor, tmp);
    for (int d = 0; d < D; ++d) out[d] = out[d] + tmp[d];
}

void bsp_fill_preorder(BspArrays &t);

// [Julia stdlib] Statistics.median!: middle order statistic, or a/2 + b/2 of the two middle ones
static double median_inplace(std::vector<double> &v)
{
    const size_t n = v.size();
    const size_t mid = n / 2;
    std::nth_element(v.begin(), v.begin() + mid, v.end());
    const double hi = v[mid];
    if (n & 1) return hi;
    const double lo = *std::max_element(v.begin(), v.begin() + mid);
    return lo / 2.0 + hi / 2.0;
}

void bsp_fill_preorder(BspArrays &t)
{
    t.pre.clear();
    t.pre.reserve((size_t)(t.P - 1));
    // iterative pre-order over the complete tree of internal depth levels-2
    std::vector<std::pair<int64_t, int>> stack;
    stack.push_back({0, 0});
    while (!stack.empty()) {
        auto [node, depth] = stack.back();
        stack.pop_back();
        if (depth == t.levels - 1) continue;
        t.pre.push_back(node);
        stack.push_back({2 * node + 2, depth + 1});
        stack.push_back({2 * node + 1, depth + 1});
    }
}

// setuppartition (partition.jl:106-129)
int bsp_build(int D, int64_t N, const double *X, int levels, int sign_mode, int dot_mode, BspArrays &t)
{
    t.D = D; t.levels = levels; t.N = N; t.dot_mode = dot_mode;
    t.P = (int64_t)1 << (levels - 1);
    t.v.assign((size_t)((t.P - 1) * D), 0.0);
    t.c.assign((size_t)(t.P - 1), 0.0);
    std::vector<int64_t> perm((size_t)N), scratch((size_t)N);
    for (int64_t i = 0; i < N; ++i) perm[(size_t)i] = i;
    std::vector<int64_t> seg_off{0, N};      // node segments of the current depth
    std::vector<double> ev, evs;
    for (int depth = 0; depth < levels - 1; ++depth) {
        const int64_t nodes = (int64_t)1 << depth;
        const int64_t heap0 = nodes - 1;
        std::vector<int64_t> next_off((size_t)(2 * nodes + 1), 0);
        for (int64_t nd = 0; nd < nodes; ++nd) {
            const int64_t b = seg_off[(size_t)nd], e = seg_off[(size_t)nd + 1], n = e - b;
            if (n <= 0) {
                set_error("BSP node %lld at depth %d has no points (N too small for levels=%d, or many duplicates)",
                          (long long)nd, depth, levels);
                return -3;
            }
            const int64_t *idx = perm.data() + b;
            double mu[MAX_D];
            pairwise_sum(D, X, idx, 0, n - 1, mu);                       // gethyperplane :89
            double *v = t.v.data() + (heap0 + nd) * D;
            bsp_direction(D, mu, n, X + idx[0] * D, sign_mode, v);      // :89-94 (x1 = first point of the node)
            ev.resize((size_t)n);
            for (int64_t i = 0; i < n; ++i) ev[(size_t)i] = dot_seq(D, v, X + idx[i] * D, dot_mode);   // splitpoints :69
            evs = ev;
            const double c = median_inplace(evs);                        // :70
            t.c[(size_t)(heap0 + nd)] = c;
            int64_t nl = 0;                                              // stable partition, left = e < c (:72-80)
            for (int64_t i = 0; i < n; ++i) nl += ev[(size_t)i] < c;
            int64_t a = b, r = b + nl;
            for (int64_t i = 0; i < n; ++i) {
                if (ev[(size_t)i] < c) scratch[(size_t)a++] = idx[i]; else scratch[(size_t)r++] = idx[i];
            }
            next_off[(size_t)(2 * nd + 1)] = b + nl;
            next_off[(size_t)(2 * nd + 2)] = e;
        }
        next_off[0] = 0;
        perm.swap(scratch);
        seg_off.swap(next_off);
    }
    t.leaf_off.assign(seg_off.begin(), seg_off.end());
    t.leaf_inds.assign(perm.begin(), perm.end());
    bsp_fill_preorder(t);
    return 0;
}

int bsp_from_hyperplanes(int D, int levels, const double *hp_v, const double *hp_c, int dot_mode, BspArrays &t)
{
    t.D = D; t.levels = levels; t.N = 0; t.dot_mode = dot_mode;
    t.P = (int64_t)1 << (levels - 1);
    t.v.assign((size_t)((t.P - 1) * D), 0.0);
    t.c.assign((size_t)(t.P - 1), 0.0);
    bsp_fill_preorder(t);
    for (int64_t k = 0; k < t.P - 1; ++k) {
        const int64_t h = t.pre[(size_t)k];
        for (int d = 0; d < D; ++d) t.v[(size_t)(h * D + d)] = hp_v[k * D + d];
        t.c[(size_t)h] = hp_c[k];
    }
    t.leaf_off.assign((size_t)(t.P + 1), 0);
    t.leaf_inds.clear();
    return 0;
}

// findpartition (partition.jl:248-262)
int64_t bsp_find(const BspArrays &t, const double *x)
{
    int64_t node = 0;
    for (int l = 1; l < t.levels; ++l)
        node = (dot_seq(t.D, t.v.data() + node * t.D, x, t.dot_mode) < t.c[(size_t)node]) ? 2 * node + 1 : 2 * node + 2;
    return node - (t.P - 1);
}

// find-eps-partitions (partition.jl:269-298) without recursion: explicit stack, left before right
static int64_t find_eps(const BspArrays &t, const double *x, double eps, int64_t *out, std::vector<int64_t> &stack)
{
    int64_t cnt = 0;
    stack.clear();
    stack.push_back(0);
    const int64_t first_leaf = t.P - 1;
    while (!stack.empty()) {
        const int64_t node = stack.back();
        stack.pop_back();
        if (node >= first_leaf) { out[cnt++] = node - first_leaf; continue; }
        const double e = dot_seq(t.D, t.v.data() + node * t.D, x, t.dot_mode);
        const double c = t.c[(size_t)node];
        if (e > c - eps) stack.push_back(2 * node + 2);     // popped second
        if (e < c + eps) stack.push_back(2 * node + 1);     // popped first
    }
    return cnt;
}

// organizetrainingsets (partition.jl:301-357)
int bsp_assign(const BspArrays &t, int64_t N, const double *X, double eps, int64_t *offsets, int64_t *inds,
               int64_t *list_offsets, int64_t *lists)
{
    std::vector<int64_t> list((size_t)t.P), stack, fill((size_t)t.P, 0);
    for (int64_t r = 0; r <= t.P; ++r) offsets[r] = 0;
    for (int64_t n = 0; n < N; ++n) {
        const int64_t cnt = find_eps(t, X + n * t.D, eps, list.data(), stack);
        for (int64_t m = 0; m < cnt; ++m) offsets[list[(size_t)m] + 1]++;
    }
    for (int64_t r = 0; r < t.P; ++r) offsets[r + 1] += offsets[r];
    if (!inds && !list_offsets && !lists) return 0;
    int64_t tot = 0;
    if (list_offsets) list_offsets[0] = 0;
    for (int64_t n = 0; n < N; ++n) {
        const int64_t cnt = find_eps(t, X + n * t.D, eps, list.data(), stack);
        for (int64_t m = 0; m < cnt; ++m) {
            const int64_t r = list[(size_t)m];
            if (inds) inds[offsets[r] + fill[(size_t)r]] = n;
            fill[(size_t)r]++;
            if (lists) lists[tot + m] = r;
        }
        tot += cnt;
        if (list_offsets) list_offsets[n + 1] = tot;
    }
    return 0;
}

// findneighbourpartitions (mixtureGP.jl:339-405)
int64_t bsp_neighbours(const BspArrays &t, const double *p, double radius, double delta, int64_t home,
                       int64_t *region_inds, double *ts, double *zs, uint8_t *keep)
{
    const int D = t.D;
    int64_t j = 0;
    double z1[MAX_D], z2[MAX_D];
    for (int64_t i = 0; i < t.P - 1; ++i) {
        const int64_t h = t.pre[(size_t)i];
        const double *u = t.v.data() + h * D;
        const double tt = -dot_seq(D, u, p, t.dot_mode) + t.c[(size_t)h];
        double s = 0.0;
        for (int d = 0; d < D; ++d) {
            const double zd = p[d] + tt * u[d];
            if (zs) zs[d + i * D] = zd;
            const double r = zd - p[d];
            s = (d == 0) ? r * r : s + r * r;
        }
        if (ts) ts[i] = tt;
        if (keep) keep[i] = 0;
        if (std::sqrt(s) < radius) {
            const double tp = tt + delta, tm = tt - delta;
            for (int d = 0; d < D; ++d) { z1[d] = p[d] + tp * u[d]; z2[d] = p[d] + tm * u[d]; }
            const int64_t r1 = bsp_find(t, z1), r2 = bsp_find(t, z2);
            if ((r2 == home) != (r1 == home)) {
                if (keep) keep[i] = 1;
                region_inds[j++] = (r1 == home) ? r2 : r1;
            }
        }
    }
    return j;
}

}  // namespace pmk
