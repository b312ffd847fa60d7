// Multi-GPU predict step inside the C ABI: the library owns an RCCL communicator (one rank per GPU, xGMI) and runs
// the request / response exchange of a sharded model on the context's stream.
//
// RCCL is bound at run time (dlopen of librccl.so.1 on the first pmk_comm_* call): libpmk_hip.so itself has no
// link-time dependency on it, single-GPU users never load it, and a host process that already carries an RCCL
// (PyTorch) shares that copy.
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <new>

#include "pmk_internal.h"

namespace {

// the handful of RCCL entry points used, by their documented C signatures (rccl/rccl.h of ROCm 7.2)
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;       // ncclSuccess == 0
enum { ncclInt32 = 2, ncclInt64 = 4, ncclFloat64 = 8 };

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

bool rccl_load()
{
    std::call_once(g_rccl_once, [] {
        void *h = nullptr;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!h) return;
        Rccl r;
        r.lib = h;
#define PMK_SYM(field, name) *(void **)(&r.field) = dlsym(h, name)
        PMK_SYM(GetUniqueId, "ncclGetUniqueId");
        PMK_SYM(CommInitRank, "ncclCommInitRank");
        PMK_SYM(CommDestroy, "ncclCommDestroy");
        PMK_SYM(GroupStart, "ncclGroupStart");
        PMK_SYM(GroupEnd, "ncclGroupEnd");
        PMK_SYM(Send, "ncclSend");
        PMK_SYM(Recv, "ncclRecv");
        PMK_SYM(AllGather, "ncclAllGather");
        PMK_SYM(GetErrorString, "ncclGetErrorString");
#undef PMK_SYM
        if (r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv &&
            r.AllGather && r.GetErrorString)
            g_rccl = r;
    });
    if (!g_rccl.lib) {
        const char *why = dlerror();          // one call: it clears the message
        pmk::set_error("RCCL is not available (librccl.so.1 could not be loaded, or lacks an entry point: %s)", why ? why : "?");
    }
    return g_rccl.lib != nullptr;
}

}  // namespace

struct pmk_comm {
    pmk_ctx *ctx = nullptr;
    int rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    // exchange workspaces on the device, grow only
    int64_t *d_counts = nullptr;          // world (mine) + world * world (table)
    double *d_xs = nullptr, *d_rx = nullptr, *d_ru = nullptr, *d_rv = nullptr;
    int32_t *d_rg = nullptr, *d_rr = nullptr;
    int64_t send_cap = 0, recv_cap = 0;
    int D_cap = 0;
    int force_exchange = 0;               // tests: run the exchange (self sends through RCCL) even at world == 1
    pmk_query *remote = nullptr;          // the received requests, as a query object that is reloaded every step
    pmk_model *remote_model = nullptr;
    int64_t remote_P = 0; int remote_D = 0;   // tree / dimension the remote query was created for (its buffers are sized by them)
    double *d_ag = nullptr; int64_t ag_cap = 0;   // all-gather variant: [2][maxc] out + [world][2][maxc] in
    int64_t bytes_sent = 0, bytes_recv = 0;   // payload of the last predict step's exchange (this rank)
};

#define PMK_NCCL(expr)                                                                               \
    do {                                                                                             \
        ncclResult_t r__ = (expr);                                                                   \
        if (r__ != 0) {                                                                              \
            pmk::set_error("%s failed at %s:%d: %s", #expr, __FILE__, __LINE__, g_rccl.GetErrorString(r__)); \
            return -101;                                                                             \
        }                                                                                            \
    } while (0)

// inside ncclGroupStart / ncclGroupEnd: a failing call must not leave the group open
#define PMK_NCCL_IN_GROUP(expr)                                                                      \
    do {                                                                                             \
        ncclResult_t r__ = (expr);                                                                   \
        if (r__ != 0) {                                                                              \
            pmk::set_error("%s failed at %s:%d: %s", #expr, __FILE__, __LINE__, g_rccl.GetErrorString(r__)); \
            (void)g_rccl.GroupEnd();                                                                 \
            return -101;                                                                             \
        }                                                                                            \
    } while (0)

using namespace pmk;

extern "C" {

int pmk_comm_unique_id(void *id_out)
{
    if (!id_out) { set_error("pmk_comm_unique_id: NULL"); return -1; }
    if (!rccl_load()) return -101;
    ncclUniqueId id;
    PMK_NCCL(g_rccl.GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return 0;
}

int pmk_comm_create(pmk_ctx *ctx, int rank, int world, const void *id, pmk_comm **out)
{
    if (!out) { set_error("pmk_comm_create: out is NULL"); return -5; }
    *out = nullptr;
    if (!ctx) { set_error("pmk_comm_create: ctx is NULL"); return -1; }
    if (world < 1 || rank < 0 || rank >= world) { set_error("pmk_comm_create: rank %d of %d", rank, world); return -2; }
    if (!id) { set_error("pmk_comm_create: id is NULL"); return -4; }
    if (!rccl_load()) return -101;
    PMK_HIP(hipSetDevice(ctx->device));
    pmk_comm *c = new (std::nothrow) pmk_comm();
    if (!c) { set_error("out of memory"); return -100; }
    c->ctx = ctx; c->rank = rank; c->world = world;
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, uid, rank);
    if (r != 0) {
        set_error("ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
        delete c;
        return -101;
    }
    if (hipMalloc((void **)&c->d_counts, sizeof(int64_t) * (size_t)(2 * (world + 1) + world * (world + 1))) != hipSuccess) {
        set_error("pmk_comm_create: out of device memory");
        g_rccl.CommDestroy(c->comm);
        delete c;
        return -100;
    }
    *out = c;
    return 0;
}

void pmk_comm_destroy(pmk_comm *c)
{
    if (!c) return;
    if (c->remote) pmk_query_destroy(c->remote);
    for (void *p : {(void *)c->d_counts, (void *)c->d_xs, (void *)c->d_rx, (void *)c->d_ru, (void *)c->d_rv, (void *)c->d_rg,
                    (void *)c->d_rr, (void *)c->d_ag})
        if (p) (void)hipFree(p);
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    delete c;
}

/* include/pmk_test.h */
int pmk_test_comm_force_exchange(pmk_comm *c, int on)
{
    if (!c) { set_error("pmk_test_comm_force_exchange: comm is NULL"); return -1; }
    c->force_exchange = on;
    return 0;
}

int pmk_comm_rank(const pmk_comm *c) { return c ? c->rank : -1; }
int pmk_comm_size(const pmk_comm *c) { return c ? c->world : -1; }

// (first, count) of every rank's segment of a region-sorted item list: rank r owns the leaves
// [r P / world, (r + 1) P / world), so its items are region_offsets[lo] .. region_offsets[hi].  Pure host arithmetic.
int pmk_shard_segments(const int64_t *region_offsets, int64_t P_global, int world, int64_t *first, int64_t *count)
{
    if (!region_offsets || !first || !count) { set_error("pmk_shard_segments: NULL argument"); return -1; }
    if (world < 1 || P_global < world || P_global % world) {
        set_error("pmk_shard_segments: the number of leaves (%lld) must be a multiple of the world size (%d)", (long long)P_global,
                  world);
        return -2;
    }
    const int64_t per = P_global / world;
    for (int r = 0; r < world; ++r) {
        first[r] = region_offsets[r * per];
        count[r] = region_offsets[(r + 1) * per] - region_offsets[r * per];
    }
    return 0;
}

static int comm_reserve(pmk_comm *c, int D, int64_t nsend, int64_t nrecv)
{
    if (nsend > c->send_cap || D > c->D_cap) {
        for (void *p : {(void *)c->d_xs, (void *)c->d_rg}) if (p) (void)hipFree(p);
        c->d_xs = nullptr; c->d_rg = nullptr; c->send_cap = 0;
        const int64_t cap = std::max<int64_t>(nsend + nsend / 8, 1024);
        PMK_HIP(hipMalloc((void **)&c->d_xs, sizeof(double) * (size_t)(cap * MAX_D)));
        PMK_HIP(hipMalloc((void **)&c->d_rg, sizeof(int32_t) * (size_t)cap));
        c->send_cap = cap;
    }
    if (nrecv > c->recv_cap || D > c->D_cap) {
        for (void *p : {(void *)c->d_rx, (void *)c->d_rr, (void *)c->d_ru, (void *)c->d_rv}) if (p) (void)hipFree(p);
        c->d_rx = nullptr; c->d_rr = nullptr; c->d_ru = nullptr; c->d_rv = nullptr; c->recv_cap = 0;
        const int64_t cap = std::max<int64_t>(nrecv + nrecv / 8, 1024);
        PMK_HIP(hipMalloc((void **)&c->d_rx, sizeof(double) * (size_t)(cap * MAX_D)));
        PMK_HIP(hipMalloc((void **)&c->d_rr, sizeof(int32_t) * (size_t)cap));
        PMK_HIP(hipMalloc((void **)&c->d_ru, sizeof(double) * (size_t)cap));
        PMK_HIP(hipMalloc((void **)&c->d_rv, sizeof(double) * (size_t)cap));
        c->recv_cap = cap;
    }
    c->D_cap = MAX_D;
    return 0;
}

// One predict step of a model whose leaves AND queries are sharded over the communicator's ranks
// (querymixtureGP!, src/RKHS/mixtureGP.jl:159-294, for this rank's queries):
//   plan of the own queries against the global tree (K5 + sort)
//   -> the (point, region) requests of every segment of the sorted item list to the rank that owns those leaves
//   -> queryinner! (K4) for everything received
//   -> (u, v) back into the requester's item buffers -> mixture (K6).
// Everything after the plan is enqueued on the context's stream (RCCL included); the host blocks only where it needs
// sizes (the plan's counts, the all-gathered segment table, the received items' region offsets).
int pmk_query_predict_sharded(pmk_query *q, pmk_comm *c, const pmk_kernel_desc *th, const pmk_kernel_desc *weight_th,
                              double radius, double delta, int64_t *total_items)
{
    if (!q || !c) { set_error("pmk_query_predict_sharded: NULL argument"); return -1; }
    pmk_model *m = q->m;
    pmk_ctx *ctx = m->ctx;
    if (ctx != c->ctx) { set_error("pmk_query_predict_sharded: the query's model lives on another context"); return -2; }
    const int W = c->world;
    if (m->P * W != m->P_global || m->leaf_base != (int64_t)c->rank * m->P) {
        set_error("pmk_query_predict_sharded: rank %d of %d must hold the leaves [%lld, %lld) of %lld, it holds [%lld, %lld)", c->rank, W,
                  (long long)(c->rank * (m->P_global / W)), (long long)((c->rank + 1) * (m->P_global / W)), (long long)m->P_global,
                  (long long)m->leaf_base, (long long)(m->leaf_base + m->P));
        return -3;
    }
    // The call is collective: a rank that fails locally must not leave its peers inside a send / receive.  Every
    // fallible local step comes before a collective, and its status travels with the next all-gather so that all ranks
    // give up together: (1) the plan, with the segment table; (2) the buffers, in a second (W words) all-gather.
    int rc = pmk_query_plan(q, radius, delta);
    if (!rc && total_items) *total_items = q->total;
    c->bytes_sent = c->bytes_recv = 0;
    if (W == 1 && !c->force_exchange) {
        if (rc) return rc;
        if ((rc = pmk_query_items(q, th))) return rc;
        return pmk_query_mix(q, weight_th, 0, q->Nq);
    }
    PMK_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int W1 = W + 1;                               // a row of the table: W segment sizes + the rank's status
    std::vector<int64_t> sfirst((size_t)W, 0), row((size_t)W1, 0), table((size_t)(W * W1));
    if (!rc) rc = pmk_shard_segments(q->roff.data(), m->P_global, W, sfirst.data(), row.data());
    row[(size_t)W] = rc;
    int64_t *d_row = c->d_counts, *d_table = c->d_counts + 2 * W1;
    PMK_HIP(hipMemcpyAsync(d_row, row.data(), sizeof(int64_t) * (size_t)W1, hipMemcpyHostToDevice, s));
    PMK_NCCL(g_rccl.AllGather(d_row, d_table, (size_t)W1, ncclInt64, c->comm, s));
    PMK_HIP(hipMemcpyAsync(table.data(), d_table, sizeof(int64_t) * (size_t)(W * W1), hipMemcpyDeviceToHost, s));
    PMK_HIP(hipStreamSynchronize(s));
    for (int r = 0; r < W; ++r)
        if (table[(size_t)(r * W1 + W)] != 0) {
            if (!rc) { set_error("pmk_query_predict_sharded: rank %d failed in its plan (%lld)", r, (long long)table[(size_t)(r * W1 + W)]); rc = -6; }
            return rc;
        }
    const std::vector<int64_t> &scount = row;
    std::vector<int64_t> rcount((size_t)W), rfirst((size_t)W);
    int64_t nrecv = 0;
    for (int r = 0; r < W; ++r) {
        rcount[(size_t)r] = table[(size_t)(r * W1 + c->rank)];      // what rank r asks of me
        rfirst[(size_t)r] = nrecv;
        nrecv += rcount[(size_t)r];
    }
    // buffers and the query object of the received requests (recreated when the tree or the dimension changed: its
    // region-offset array is sized by the tree it was created for)
    if (nrecv > 0x7fffffff) { set_error("pmk_query_predict_sharded: too many requests (%lld)", (long long)nrecv); rc = -5; }
    if (!rc) rc = comm_reserve(c, m->D, q->total, nrecv);
    if (!rc && (!c->remote || c->remote_model != m || c->remote_P != m->P_global || c->remote_D != m->D)) {
        if (c->remote) pmk_query_destroy(c->remote);
        c->remote = nullptr;
        rc = pmk_query_create(m, 0, nullptr, &c->remote);
        c->remote_model = m; c->remote_P = m->P_global; c->remote_D = m->D;
    }
    {
        int64_t st = rc, *d_st = c->d_counts + W1;              // second agreement: W status words
        std::vector<int64_t> all((size_t)W, 0);
        PMK_HIP(hipMemcpyAsync(d_st, &st, sizeof(int64_t), hipMemcpyHostToDevice, s));
        PMK_NCCL(g_rccl.AllGather(d_st, d_table, 1, ncclInt64, c->comm, s));
        PMK_HIP(hipMemcpyAsync(all.data(), d_table, sizeof(int64_t) * (size_t)W, hipMemcpyDeviceToHost, s));
        PMK_HIP(hipStreamSynchronize(s));
        for (int r = 0; r < W; ++r)
            if (all[(size_t)r] != 0) {
                if (!rc) { set_error("pmk_query_predict_sharded: rank %d could not set up its buffers (%lld)", r, (long long)all[(size_t)r]); rc = -6; }
                return rc;
            }
    }
    const int D = m->D;
    // 1. requests out, requests in
    if (q->total > 0 && (rc = launch_export_requests(q, 0, q->total, c->d_xs, c->d_rg, s))) return rc;
    PMK_NCCL(g_rccl.GroupStart());
    for (int r = 0; r < W; ++r) {
        if (scount[(size_t)r] > 0) {
            PMK_NCCL_IN_GROUP(g_rccl.Send(c->d_xs + sfirst[(size_t)r] * D, (size_t)(scount[(size_t)r] * D), ncclFloat64, r, c->comm, s));
            PMK_NCCL_IN_GROUP(g_rccl.Send(c->d_rg + sfirst[(size_t)r], (size_t)scount[(size_t)r], ncclInt32, r, c->comm, s));
        }
        if (rcount[(size_t)r] > 0) {
            PMK_NCCL_IN_GROUP(g_rccl.Recv(c->d_rx + rfirst[(size_t)r] * D, (size_t)(rcount[(size_t)r] * D), ncclFloat64, r, c->comm, s));
            PMK_NCCL_IN_GROUP(g_rccl.Recv(c->d_rr + rfirst[(size_t)r], (size_t)rcount[(size_t)r], ncclInt32, r, c->comm, s));
        }
        if (r != c->rank) {
            c->bytes_sent += scount[(size_t)r] * (8 * D + 4) + rcount[(size_t)r] * 16;
            c->bytes_recv += rcount[(size_t)r] * (8 * D + 4) + scount[(size_t)r] * 16;
        }
    }
    PMK_NCCL(g_rccl.GroupEnd());
    // 2. queryinner! for everything received
    if ((rc = query_set_items(c->remote, nrecv, c->d_rx, c->d_rr))) return rc;
    if ((rc = pmk_query_items(c->remote, th))) return rc;
    if (nrecv > 0 && (rc = launch_export_results(c->remote, c->d_ru, c->d_rv, s))) return rc;
    // 3. results back, straight into the requester's sorted item buffers
    PMK_NCCL(g_rccl.GroupStart());
    for (int r = 0; r < W; ++r) {
        if (rcount[(size_t)r] > 0) {
            PMK_NCCL_IN_GROUP(g_rccl.Send(c->d_ru + rfirst[(size_t)r], (size_t)rcount[(size_t)r], ncclFloat64, r, c->comm, s));
            PMK_NCCL_IN_GROUP(g_rccl.Send(c->d_rv + rfirst[(size_t)r], (size_t)rcount[(size_t)r], ncclFloat64, r, c->comm, s));
        }
        if (scount[(size_t)r] > 0) {
            PMK_NCCL_IN_GROUP(g_rccl.Recv(q->d_u + sfirst[(size_t)r], (size_t)scount[(size_t)r], ncclFloat64, r, c->comm, s));
            PMK_NCCL_IN_GROUP(g_rccl.Recv(q->d_v + sfirst[(size_t)r], (size_t)scount[(size_t)r], ncclFloat64, r, c->comm, s));
        }
    }
    PMK_NCCL(g_rccl.GroupEnd());
    // 4. blend
    return pmk_query_mix(q, weight_th, 0, q->Nq);
}

// The north star's literal form of the same step: the QUERIES ARE REPLICATED (every rank passes all of them and plans all
// of them against the global tree), every rank evaluates queryinner! for the items that fall into ITS leaves, and ONE
// ncclAllGather of equal, padded (u, v) slices -- every rank knows every segment's size from its own plan, so no table
// is exchanged -- completes every rank's item buffers before the mixture weights are applied to all queries.
// Against the request / response form above: no point travels, every (u, v) travels to every rank ((W - 1) x 16 B per
// item in, against 36 B per item that crosses ranks), and the plan is W times larger per rank.
int pmk_query_predict_allgather(pmk_query *q, pmk_comm *c, const pmk_kernel_desc *th, const pmk_kernel_desc *weight_th,
                                double radius, double delta, int64_t *total_items)
{
    if (!q || !c) { set_error("pmk_query_predict_allgather: NULL argument"); return -1; }
    pmk_model *m = q->m;
    pmk_ctx *ctx = m->ctx;
    if (ctx != c->ctx) { set_error("pmk_query_predict_allgather: the query's model lives on another context"); return -2; }
    const int W = c->world;
    if (m->P * W != m->P_global || m->leaf_base != (int64_t)c->rank * m->P) {
        set_error("pmk_query_predict_allgather: rank %d of %d holds the leaves [%lld, %lld) of %lld", c->rank, W,
                  (long long)m->leaf_base, (long long)(m->leaf_base + m->P), (long long)m->P_global);
        return -3;
    }
    // the plan is a function of (queries, tree) alone: identical on every rank, so a failure is one on every rank
    int rc = pmk_query_plan(q, radius, delta);
    if (rc) return rc;
    if (total_items) *total_items = q->total;
    c->bytes_sent = c->bytes_recv = 0;
    if ((rc = pmk_query_items(q, th))) return rc;                 // the items of this rank's leaves
    if (W > 1 || c->force_exchange) {
        PMK_HIP(hipSetDevice(ctx->device));
        hipStream_t s = ctx->stream;
        std::vector<int64_t> sfirst((size_t)W), scount((size_t)W);
        if ((rc = pmk_shard_segments(q->roff.data(), m->P_global, W, sfirst.data(), scount.data()))) return rc;
        const int64_t maxc = std::max<int64_t>(1, *std::max_element(scount.begin(), scount.end()));
        if ((int64_t)(W + 1) * 2 * maxc > c->ag_cap) {           // deterministic in the plan: every rank grows alike
            if (c->d_ag) (void)hipFree(c->d_ag);
            c->d_ag = nullptr; c->ag_cap = 0;
            const int64_t cap = (int64_t)(W + 1) * 2 * (maxc + maxc / 8);
            PMK_HIP(hipMalloc((void **)&c->d_ag, sizeof(double) * (size_t)cap));
            c->ag_cap = cap;
        }
        double *out = c->d_ag, *in = c->d_ag + 2 * maxc;
        const int64_t mine = scount[(size_t)c->rank];
        if (mine > 0) {
            PMK_HIP(hipMemcpyAsync(out, q->d_u + sfirst[(size_t)c->rank], sizeof(double) * (size_t)mine, hipMemcpyDeviceToDevice, s));
            PMK_HIP(hipMemcpyAsync(out + maxc, q->d_v + sfirst[(size_t)c->rank], sizeof(double) * (size_t)mine, hipMemcpyDeviceToDevice, s));
        }
        PMK_NCCL(g_rccl.AllGather(out, in, (size_t)(2 * maxc), ncclFloat64, c->comm, s));
        for (int r = 0; r < W; ++r) {
            if (r == c->rank && !c->force_exchange) continue;
            if (scount[(size_t)r] > 0) {
                PMK_HIP(hipMemcpyAsync(q->d_u + sfirst[(size_t)r], in + (int64_t)r * 2 * maxc, sizeof(double) * (size_t)scount[(size_t)r], hipMemcpyDeviceToDevice, s));
                PMK_HIP(hipMemcpyAsync(q->d_v + sfirst[(size_t)r], in + (int64_t)r * 2 * maxc + maxc, sizeof(double) * (size_t)scount[(size_t)r], hipMemcpyDeviceToDevice, s));
            }
        }
        c->bytes_sent = (int64_t)(W - 1) * 2 * maxc * 8;
        c->bytes_recv = (int64_t)(W - 1) * 2 * maxc * 8;
    }
    return pmk_query_mix(q, weight_th, 0, q->Nq);
}

// payload bytes this rank sent / received in the exchange of its last predict step (either form)
int pmk_comm_last_bytes(const pmk_comm *c, int64_t *sent, int64_t *received)
{
    if (!c) { set_error("pmk_comm_last_bytes: comm is NULL"); return -1; }
    if (sent) *sent = c->bytes_sent;
    if (received) *received = c->bytes_recv;
    return 0;
}

}  // extern "C"
