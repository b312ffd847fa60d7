// Device-resident BSP build: setuppartition (src/patchwork/partition.jl:106-129 with gethyperplane :86-100,
// splitpoints :64-83, createchildren :166-217) level by level on the GPU, bit-identical to the host build of
// pmk_bsp.cpp (tests/test_gpu_parity.py compares the two).
//
// Layout: the points of every node of the current depth are one contiguous segment of Xp (coordinates in node
// order, point-major), perm (original index) and node_of (node id within the depth).  Per depth:
//   1. coordinate sums with the reference's pairwise order: the <= 1024-point sequential leaf blocks of the
//      recursion are summed one thread per block; the few partial sums are combined on the host in recursion
//      order, which also derives the unit normal (tiny scalar work kept in the contraction-free host file),
//   2. e = v . x per point (sequential products, no FMA),
//   3. median: radix sort of e, then a stable radix sort by node id -> e sorted inside every segment; the middle
//      order statistic(s) give c exactly as Statistics.median does (a/2 + b/2),
//   4. stable in-segment partition by e < c: flags -> one exclusive scan -> scatter of (perm, Xp, node_of).
// All steps are order-deterministic, so the integer outputs (leaf index lists) match the host's exactly.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <vector>

#include "pmk_device.h"

namespace pmk {

void bsp_direction(int D, const double *sum, int64_t n, const double *x1, int sign_mode, double *v);
void bsp_pairwise_blocks(int64_t first, int64_t last, int64_t base, std::vector<int64_t> &blk_first,
                         std::vector<int64_t> &blk_last);
void bsp_pairwise_combine(int D, int64_t first, int64_t last, const double *partials, int64_t &cursor, double *out);
void bsp_fill_preorder(BspArrays &t);

namespace {

template <int D>
__global__ __launch_bounds__(64) void block_sum_kernel(const double *__restrict__ Xp, const int64_t *__restrict__ blk_first,
                                                       const int64_t *__restrict__ blk_last, int64_t nblk,
                                                       double *__restrict__ partial)
{
    const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= nblk) return;
    const int64_t f = blk_first[t], l = blk_last[t];
    double s[D];
#pragma unroll
    for (int d = 0; d < D; ++d) s[d] = Xp[f * D + d];
    for (int64_t i = f + 1; i <= l; ++i) {
#pragma unroll
        for (int d = 0; d < D; ++d) s[d] = s[d] + Xp[i * D + d];
    }
#pragma unroll
    for (int d = 0; d < D; ++d) partial[t * D + d] = s[d];
}

template <int D>
__global__ void first_points_kernel(const double *__restrict__ Xp, const int64_t *__restrict__ seg_off, int64_t nodes,
                                    int64_t N, double *__restrict__ first)
{
    const int64_t nd = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (nd >= nodes) return;
    const int64_t b = seg_off[nd];
#pragma unroll
    for (int d = 0; d < D; ++d) first[nd * D + d] = (b < N) ? Xp[b * D + d] : 0.0;
}

template <int D>
__global__ void project_kernel(int64_t N, const double *__restrict__ Xp, const int32_t *__restrict__ node_of,
                               const double *__restrict__ v, double *__restrict__ e, int dot_mode)
{
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const double *u = v + (int64_t)node_of[i] * D;
    double x[D];
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = Xp[i * D + d];
    e[i] = dot_seq<D>(u, x, dot_mode);
}

// c per node from the in-segment sorted projections (Statistics.median: a/2 + b/2 for even counts)
__global__ void median_kernel(int64_t nodes, const int64_t *__restrict__ seg_off, const double *__restrict__ es,
                              double *__restrict__ c)
{
    const int64_t nd = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (nd >= nodes) return;
    const int64_t b = seg_off[nd], n = seg_off[nd + 1] - b;
    if (n <= 0) { c[nd] = 0.0; return; }
    const int64_t mid = n / 2;
    const double hi = es[b + mid];
    c[nd] = (n & 1) ? hi : es[b + mid - 1] / 2.0 + hi / 2.0;
}

__global__ void flag_kernel(int64_t N, const double *__restrict__ e, const int32_t *__restrict__ node_of,
                            const double *__restrict__ c, int32_t *__restrict__ flag)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i > N) return;
    flag[i] = (i < N && e[i] < c[node_of[i]]) ? 1 : 0;      // flag[N] = 0 closes the scan
}

template <int D>
__global__ void split_kernel(int64_t N, const int64_t *__restrict__ seg_off, const int32_t *__restrict__ flag,
                             const int32_t *__restrict__ L, const int32_t *__restrict__ node_of,
                             const int32_t *__restrict__ perm, const double *__restrict__ Xp,
                             int32_t *__restrict__ node2, int32_t *__restrict__ perm2, double *__restrict__ Xp2)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int32_t nd = node_of[i];
    const int64_t b = seg_off[nd], e_ = seg_off[nd + 1];
    const int64_t Lb = L[b], nl = L[e_] - Lb, r = L[i] - Lb;
    const bool left = flag[i] != 0;
    const int64_t dest = left ? b + r : b + nl + (i - b - r);
    node2[dest] = 2 * nd + (left ? 0 : 1);
    perm2[dest] = perm[i];
#pragma unroll
    for (int d = 0; d < D; ++d) Xp2[dest * D + d] = Xp[i * D + d];
}

__global__ void next_offsets_kernel(int64_t nodes, int64_t N, const int64_t *__restrict__ seg_off,
                                    const int32_t *__restrict__ L, int64_t *__restrict__ next_off)
{
    const int64_t nd = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (nd > nodes) return;
    if (nd == nodes) { next_off[2 * nodes] = N; return; }
    const int64_t b = seg_off[nd], e_ = seg_off[nd + 1];
    next_off[2 * nd] = b;
    next_off[2 * nd + 1] = b + (L[e_] - L[b]);
}

__global__ void init_kernel(int64_t N, int32_t *__restrict__ perm, int32_t *__restrict__ node_of)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    perm[i] = (int32_t)i;
    node_of[i] = 0;
}

struct DevBuf {
    std::vector<void *> all;
    template <typename T>
    T *get(size_t count)
    {
        void *p = nullptr;
        if (hipMalloc(&p, sizeof(T) * (count ? count : 1)) != hipSuccess) return nullptr;
        all.push_back(p);
        return static_cast<T *>(p);
    }
    ~DevBuf()
    {
        for (void *p : all) (void)hipFree(p);
    }
};

inline unsigned blocks_for(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

template <int D>
int build_levels(pmk_ctx *c, int64_t N, const double *X, int levels, int sign_mode, int dot_mode, BspArrays &t)
{
    hipStream_t s = c->stream;
    const int64_t P = t.P;
    DevBuf mem;
    double *Xp = mem.get<double>((size_t)(N * D)), *Xp2 = mem.get<double>((size_t)(N * D));
    int32_t *perm = mem.get<int32_t>((size_t)N), *perm2 = mem.get<int32_t>((size_t)N);
    int32_t *node = mem.get<int32_t>((size_t)N), *node2 = mem.get<int32_t>((size_t)N), *node_s = mem.get<int32_t>((size_t)N);
    double *e = mem.get<double>((size_t)N), *e1 = mem.get<double>((size_t)N), *e2 = mem.get<double>((size_t)N);
    int32_t *flag = mem.get<int32_t>((size_t)N + 1), *L = mem.get<int32_t>((size_t)N + 1);
    int64_t *d_seg = mem.get<int64_t>((size_t)P + 1), *d_next = mem.get<int64_t>((size_t)P + 1);
    double *d_v = mem.get<double>((size_t)(P * D)), *d_c = mem.get<double>((size_t)P), *d_first = mem.get<double>((size_t)(P * D));
    // at most one leaf block per 513 points plus one per node
    const size_t max_blk = (size_t)(N / 512 + P + 8);
    int64_t *d_bf = mem.get<int64_t>(max_blk), *d_bl = mem.get<int64_t>(max_blk);
    double *d_part = mem.get<double>(max_blk * D);
    if (!Xp || !Xp2 || !perm || !perm2 || !node || !node2 || !node_s || !e || !e1 || !e2 || !flag || !L || !d_seg ||
        !d_next || !d_v || !d_c || !d_first || !d_bf || !d_bl || !d_part) {
        set_error("pmk_bsp_build_device: out of device memory (N=%lld)", (long long)N);
        return -100;
    }
    // sort / scan scratch, sized once for the largest request
    size_t need_sort1 = 0, need_sort2 = 0, need_scan = 0;
    PMK_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, need_sort1, e, e1, node, node_s, (int)N, 0, 64, s));
    PMK_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, need_sort2, node_s, node2, e1, e2, (int)N, 0, 32, s));
    PMK_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, need_scan, flag, L, (int)(N + 1), s));
    size_t tmp_bytes = std::max(need_sort1, std::max(need_sort2, need_scan));
    void *d_tmp = mem.get<char>(tmp_bytes);
    if (!d_tmp) { set_error("pmk_bsp_build_device: out of device memory (scratch)"); return -100; }

    PMK_HIP(hipMemcpyAsync(Xp, X, sizeof(double) * (size_t)(N * D), hipMemcpyDefault, s));
    hipLaunchKernelGGL(init_kernel, dim3(blocks_for(N, 256)), dim3(256), 0, s, N, perm, node);

    std::vector<int64_t> seg_off{0, N}, next_off, blk_first, blk_last;
    std::vector<double> partials, firsts, vlevel, clevel;
    for (int depth = 0; depth < levels - 1; ++depth) {
        const int64_t nodes = (int64_t)1 << depth, heap0 = nodes - 1;
        // ---- 1. pairwise coordinate sums
        blk_first.clear(); blk_last.clear();
        for (int64_t nd = 0; nd < nodes; ++nd) {
            const int64_t b = seg_off[(size_t)nd], n = seg_off[(size_t)nd + 1] - b;
            if (n <= 0) {
                set_error("BSP node %lld at depth %d has no points (N too small for levels=%d, or many duplicates)",
                          (long long)nd, depth, levels);
                return -3;
            }
            bsp_pairwise_blocks(0, n - 1, b, blk_first, blk_last);
        }
        const int64_t nblk = (int64_t)blk_first.size();
        if ((size_t)nblk > max_blk) { set_error("pmk_bsp_build_device: internal block count overflow"); return -100; }
        PMK_HIP(hipMemcpyAsync(d_bf, blk_first.data(), sizeof(int64_t) * (size_t)nblk, hipMemcpyHostToDevice, s));
        PMK_HIP(hipMemcpyAsync(d_bl, blk_last.data(), sizeof(int64_t) * (size_t)nblk, hipMemcpyHostToDevice, s));
        PMK_HIP(hipMemcpyAsync(d_seg, seg_off.data(), sizeof(int64_t) * (size_t)(nodes + 1), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL((block_sum_kernel<D>), dim3(blocks_for(nblk, 64)), dim3(64), 0, s, Xp, d_bf, d_bl, nblk, d_part);
        hipLaunchKernelGGL((first_points_kernel<D>), dim3(blocks_for(nodes, 256)), dim3(256), 0, s, Xp, d_seg, nodes, N, d_first);
        partials.resize((size_t)(nblk * D));
        firsts.resize((size_t)(nodes * D));
        PMK_HIP(hipMemcpyAsync(partials.data(), d_part, sizeof(double) * partials.size(), hipMemcpyDeviceToHost, s));
        PMK_HIP(hipMemcpyAsync(firsts.data(), d_first, sizeof(double) * firsts.size(), hipMemcpyDeviceToHost, s));
        PMK_HIP(hipStreamSynchronize(s));
        vlevel.resize((size_t)(nodes * D));
        int64_t cursor = 0;
        for (int64_t nd = 0; nd < nodes; ++nd) {
            const int64_t n = seg_off[(size_t)nd + 1] - seg_off[(size_t)nd];
            double sum[MAX_D];
            bsp_pairwise_combine(D, 0, n - 1, partials.data(), cursor, sum);
            bsp_direction(D, sum, n, firsts.data() + nd * D, sign_mode, vlevel.data() + nd * D);
            for (int d = 0; d < D; ++d) t.v[(size_t)((heap0 + nd) * D + d)] = vlevel[(size_t)(nd * D + d)];
        }
        PMK_HIP(hipMemcpyAsync(d_v, vlevel.data(), sizeof(double) * vlevel.size(), hipMemcpyHostToDevice, s));
        // ---- 2. projections
        hipLaunchKernelGGL((project_kernel<D>), dim3(blocks_for(N, 256)), dim3(256), 0, s, N, Xp, node, d_v, e, dot_mode);
        // ---- 3. medians: sort by e, then stably by node id
        const double *sorted = e1;
        PMK_HIP(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, e, e1, node, node_s, (int)N, 0, 64, s));
        if (depth > 0) {
            PMK_HIP(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, node_s, node2, e1, e2, (int)N, 0, depth, s));
            sorted = e2;
        }
        hipLaunchKernelGGL(median_kernel, dim3(blocks_for(nodes, 256)), dim3(256), 0, s, nodes, d_seg, sorted, d_c);
        // ---- 4. stable split
        hipLaunchKernelGGL(flag_kernel, dim3(blocks_for(N + 1, 256)), dim3(256), 0, s, N, e, node, d_c, flag);
        PMK_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, flag, L, (int)(N + 1), s));
        hipLaunchKernelGGL((split_kernel<D>), dim3(blocks_for(N, 256)), dim3(256), 0, s, N, d_seg, flag, L, node, perm, Xp,
                           node2, perm2, Xp2);
        hipLaunchKernelGGL(next_offsets_kernel, dim3(blocks_for(nodes + 1, 256)), dim3(256), 0, s, nodes, N, d_seg, L, d_next);
        PMK_HIP(hipGetLastError());
        next_off.resize((size_t)(2 * nodes + 1));
        clevel.resize((size_t)nodes);
        PMK_HIP(hipMemcpyAsync(next_off.data(), d_next, sizeof(int64_t) * next_off.size(), hipMemcpyDeviceToHost, s));
        PMK_HIP(hipMemcpyAsync(clevel.data(), d_c, sizeof(double) * clevel.size(), hipMemcpyDeviceToHost, s));
        PMK_HIP(hipStreamSynchronize(s));
        for (int64_t nd = 0; nd < nodes; ++nd) t.c[(size_t)(heap0 + nd)] = clevel[(size_t)nd];
        seg_off.swap(next_off);
        std::swap(Xp, Xp2);
        std::swap(perm, perm2);
        std::swap(node, node2);
    }
    t.leaf_off.assign(seg_off.begin(), seg_off.end());
    std::vector<int32_t> hperm((size_t)N);
    PMK_HIP(hipMemcpyAsync(hperm.data(), perm, sizeof(int32_t) * (size_t)N, hipMemcpyDeviceToHost, s));
    PMK_HIP(hipStreamSynchronize(s));
    t.leaf_inds.assign(hperm.begin(), hperm.end());
    return 0;
}

// find-eps-partitions (partition.jl:269-298) for every point: depth-first, left before right, so a point's leaves
// come out in ascending order.  FILL = false counts (per point and per leaf), FILL = true writes the lists.
template <int D, bool FILL>
__global__ void eps_walk_kernel(int64_t N, const double *__restrict__ X, const double *__restrict__ v,
                                const double *__restrict__ c, int64_t P, double eps, int32_t *__restrict__ cnt,
                                unsigned long long *__restrict__ leaf_cnt, const int64_t *__restrict__ loff,
                                int32_t *__restrict__ lists, int32_t *__restrict__ pair_point, int dot_mode)
{
#pragma clang fp contract(off)
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    double x[D];
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = X[n * D + d];
    const int64_t first_leaf = P - 1;
    int64_t stack[34];
    int sp = 0;
    stack[sp++] = 0;
    int32_t found = 0;
    const int64_t base = FILL ? loff[n] : 0;
    while (sp > 0) {
        const int64_t node = stack[--sp];
        if (node >= first_leaf) {
            const int64_t leaf = node - first_leaf;
            if (FILL) {
                lists[base + found] = (int32_t)leaf;
                pair_point[base + found] = (int32_t)n;
            } else {
                atomicAdd(leaf_cnt + leaf, 1ULL);
            }
            ++found;
            continue;
        }
        const double *u = v + node * D;
        const double e = dot_seq<D>(u, x, dot_mode);
        const double cc = c[node];
        if (e > cc - eps) stack[sp++] = 2 * node + 2;     // popped second
        if (e < cc + eps) stack[sp++] = 2 * node + 1;     // popped first
    }
    if (!FILL) cnt[n] = found;
}

template <int D>
int assign_levels(pmk_ctx *c, const BspArrays &t, int64_t N, const double *X, double eps, int64_t *offsets, int64_t *inds,
                  int64_t *list_offsets, int64_t *lists)
{
    hipStream_t s = c->stream;
    const int64_t P = t.P;
    DevBuf mem;
    double *dX = mem.get<double>((size_t)(N * D));
    double *d_v = mem.get<double>((size_t)((P - 1) * D)), *d_c = mem.get<double>((size_t)(P - 1));
    int32_t *cnt = mem.get<int32_t>((size_t)N + 1);
    int64_t *loff = mem.get<int64_t>((size_t)N + 1);
    unsigned long long *leaf_cnt = mem.get<unsigned long long>((size_t)P);
    if (!dX || !d_v || !d_c || !cnt || !loff || !leaf_cnt) { set_error("pmk_bsp_assign_device: out of device memory"); return -100; }
    PMK_HIP(hipMemcpyAsync(dX, X, sizeof(double) * (size_t)(N * D), hipMemcpyDefault, s));
    PMK_HIP(hipMemcpyAsync(d_v, t.v.data(), sizeof(double) * t.v.size(), hipMemcpyHostToDevice, s));
    PMK_HIP(hipMemcpyAsync(d_c, t.c.data(), sizeof(double) * t.c.size(), hipMemcpyHostToDevice, s));
    PMK_HIP(hipMemsetAsync(leaf_cnt, 0, sizeof(unsigned long long) * (size_t)P, s));
    PMK_HIP(hipMemsetAsync(cnt, 0, sizeof(int32_t) * (size_t)(N + 1), s));
    hipLaunchKernelGGL((eps_walk_kernel<D, false>), dim3(blocks_for(N, 256)), dim3(256), 0, s, N, dX, d_v, d_c, P, eps, cnt,
                       leaf_cnt, nullptr, nullptr, nullptr, t.dot_mode);
    std::vector<unsigned long long> hcnt((size_t)P);
    PMK_HIP(hipMemcpyAsync(hcnt.data(), leaf_cnt, sizeof(unsigned long long) * (size_t)P, hipMemcpyDeviceToHost, s));
    PMK_HIP(hipStreamSynchronize(s));
    offsets[0] = 0;
    for (int64_t r = 0; r < P; ++r) offsets[r + 1] = offsets[r] + (int64_t)hcnt[(size_t)r];
    if (!inds && !list_offsets && !lists) return 0;
    const int64_t total = offsets[P];
    if (total >= 0x7fffffff) { set_error("pmk_bsp_assign_device: too many (point, leaf) pairs"); return -5; }
    // per-point offsets, then the lists and the (leaf, point) pairs in point order
    size_t need_scan = 0, need_sort = 0;
    int32_t *d_lists = mem.get<int32_t>((size_t)total), *d_pt = mem.get<int32_t>((size_t)total);
    int32_t *d_keys = mem.get<int32_t>((size_t)total), *d_sorted_pt = mem.get<int32_t>((size_t)total);
    int bits = 1;
    while (((int64_t)1 << bits) < P) ++bits;
    PMK_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, need_scan, cnt, loff, (int)(N + 1), s));
    PMK_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, need_sort, d_lists, d_keys, d_pt, d_sorted_pt, (int)total, 0, bits, s));
    const size_t tmp_bytes = std::max(need_scan, need_sort);
    void *d_tmp = mem.get<char>(tmp_bytes);
    if (!d_lists || !d_pt || !d_keys || !d_sorted_pt || !d_tmp) { set_error("pmk_bsp_assign_device: out of device memory"); return -100; }
    size_t tb = tmp_bytes;
    PMK_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp, tb, cnt, loff, (int)(N + 1), s));
    hipLaunchKernelGGL((eps_walk_kernel<D, true>), dim3(blocks_for(N, 256)), dim3(256), 0, s, N, dX, d_v, d_c, P, eps, cnt,
                       leaf_cnt, loff, d_lists, d_pt, t.dot_mode);
    PMK_HIP(hipGetLastError());
    std::vector<int32_t> h32((size_t)std::max<int64_t>(total, 1));
    if (inds && total > 0) {
        // stable sort by leaf: within a leaf the points stay in ascending order (X_set_inds, partition.jl:320-330)
        tb = tmp_bytes;
        PMK_HIP(hipcub::DeviceRadixSort::SortPairs(d_tmp, tb, d_lists, d_keys, d_pt, d_sorted_pt, (int)total, 0, bits, s));
        PMK_HIP(hipMemcpyAsync(h32.data(), d_sorted_pt, sizeof(int32_t) * (size_t)total, hipMemcpyDeviceToHost, s));
        PMK_HIP(hipStreamSynchronize(s));
        for (int64_t i = 0; i < total; ++i) inds[i] = h32[(size_t)i];
    }
    if (lists && total > 0) {
        PMK_HIP(hipMemcpyAsync(h32.data(), d_lists, sizeof(int32_t) * (size_t)total, hipMemcpyDeviceToHost, s));
        PMK_HIP(hipStreamSynchronize(s));
        for (int64_t i = 0; i < total; ++i) lists[i] = h32[(size_t)i];
    }
    if (list_offsets) {
        PMK_HIP(hipMemcpyAsync(list_offsets, loff, sizeof(int64_t) * (size_t)(N + 1), hipMemcpyDeviceToHost, s));
        PMK_HIP(hipStreamSynchronize(s));
    }
    return 0;
}

}  // namespace

// organizetrainingsets (partition.jl:301-357) on the GPU; same outputs as bsp_assign of pmk_bsp.cpp
int bsp_assign_device(pmk_ctx *c, const BspArrays &t, int64_t N, const double *X, double eps, int64_t *offsets,
                      int64_t *inds, int64_t *list_offsets, int64_t *lists)
{
    if (N == 0) {
        for (int64_t r = 0; r <= t.P; ++r) offsets[r] = 0;
        if (list_offsets) list_offsets[0] = 0;
        return 0;
    }
    switch (t.D) {
    case 1: return assign_levels<1>(c, t, N, X, eps, offsets, inds, list_offsets, lists);
    case 2: return assign_levels<2>(c, t, N, X, eps, offsets, inds, list_offsets, lists);
    case 3: return assign_levels<3>(c, t, N, X, eps, offsets, inds, list_offsets, lists);
    case 4: return assign_levels<4>(c, t, N, X, eps, offsets, inds, list_offsets, lists);
    default: set_error("pmk_bsp_assign_device: D=%d outside 1..%d", t.D, MAX_D); return -1;
    }
}

int bsp_build_device(pmk_ctx *c, int D, int64_t N, const double *X, int levels, int sign_mode, int dot_mode, BspArrays &t)
{
    t.D = D; t.levels = levels; t.N = N; t.dot_mode = dot_mode;
    t.P = (int64_t)1 << (levels - 1);
    t.v.assign((size_t)((t.P - 1) * D), 0.0);
    t.c.assign((size_t)(t.P - 1), 0.0);
    int rc;
    switch (D) {
    case 1: rc = build_levels<1>(c, N, X, levels, sign_mode, dot_mode, t); break;
    case 2: rc = build_levels<2>(c, N, X, levels, sign_mode, dot_mode, t); break;
    case 3: rc = build_levels<3>(c, N, X, levels, sign_mode, dot_mode, t); break;
    case 4: rc = build_levels<4>(c, N, X, levels, sign_mode, dot_mode, t); break;
    default: set_error("pmk_bsp_build_device: D=%d outside 1..%d", D, MAX_D); return -1;
    }
    if (rc) return rc;
    bsp_fill_preorder(t);
    return 0;
}

}  // namespace pmk
