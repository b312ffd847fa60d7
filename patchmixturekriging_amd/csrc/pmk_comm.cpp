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
    if (!g_rccl.lib) pmk::set_error("RCCL is not available (librccl.so.1 could not be loaded: %s)", dlerror() ? dlerror() : "?");
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
};

#define PMK_NCCL(expr)                                                                               \
    do {                                                                                             \
        ncclResult_t r__ = (expr);                                                                   \
        if (r__ != 0) {                                                                              \
            pmk::set_error("%s failed at %s:%d: %s", #expr, __FILE__, __LINE__, g_rccl.GetErrorString(r__)); \
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
    if (hipMalloc((void **)&c->d_counts, sizeof(int64_t) * (size_t)(world + world * world)) != hipSuccess) {
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
                    (void *)c->d_rr})
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
    int rc = pmk_query_plan(q, radius, delta);
    if (rc) return rc;
    if (total_items) *total_items = q->total;
    if (W == 1 && !c->force_exchange) {
        if ((rc = pmk_query_items(q, th))) return rc;
        return pmk_query_mix(q, weight_th, 0, q->Nq);
    }
    PMK_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    std::vector<int64_t> sfirst((size_t)W), scount((size_t)W), table((size_t)(W * W));
    if ((rc = pmk_shard_segments(q->roff.data(), m->P_global, W, sfirst.data(), scount.data()))) return rc;
    // every rank's segment sizes to every rank (W x W table, a few hundred bytes)
    PMK_HIP(hipMemcpyAsync(c->d_counts, scount.data(), sizeof(int64_t) * (size_t)W, hipMemcpyHostToDevice, s));
    PMK_NCCL(g_rccl.AllGather(c->d_counts, c->d_counts + W, (size_t)W, ncclInt64, c->comm, s));
    PMK_HIP(hipMemcpyAsync(table.data(), c->d_counts + W, sizeof(int64_t) * (size_t)(W * W), hipMemcpyDeviceToHost, s));
    PMK_HIP(hipStreamSynchronize(s));
    std::vector<int64_t> rcount((size_t)W), rfirst((size_t)W);
    int64_t nrecv = 0;
    for (int r = 0; r < W; ++r) {
        rcount[(size_t)r] = table[(size_t)(r * W + c->rank)];      // what rank r asks of me
        rfirst[(size_t)r] = nrecv;
        nrecv += rcount[(size_t)r];
    }
    if (nrecv > 0x7fffffff) { set_error("pmk_query_predict_sharded: too many requests (%lld)", (long long)nrecv); return -5; }
    if ((rc = comm_reserve(c, m->D, q->total, nrecv))) return rc;
    const int D = m->D;
    // 1. requests out, requests in
    if (q->total > 0 && (rc = launch_export_requests(q, 0, q->total, c->d_xs, c->d_rg, s))) return rc;
    PMK_NCCL(g_rccl.GroupStart());
    for (int r = 0; r < W; ++r) {
        if (scount[(size_t)r] > 0) {
            PMK_NCCL(g_rccl.Send(c->d_xs + sfirst[(size_t)r] * D, (size_t)(scount[(size_t)r] * D), ncclFloat64, r, c->comm, s));
            PMK_NCCL(g_rccl.Send(c->d_rg + sfirst[(size_t)r], (size_t)scount[(size_t)r], ncclInt32, r, c->comm, s));
        }
        if (rcount[(size_t)r] > 0) {
            PMK_NCCL(g_rccl.Recv(c->d_rx + rfirst[(size_t)r] * D, (size_t)(rcount[(size_t)r] * D), ncclFloat64, r, c->comm, s));
            PMK_NCCL(g_rccl.Recv(c->d_rr + rfirst[(size_t)r], (size_t)rcount[(size_t)r], ncclInt32, r, c->comm, s));
        }
    }
    PMK_NCCL(g_rccl.GroupEnd());
    // 2. queryinner! for everything received
    if (!c->remote || c->remote_model != m) {
        if (c->remote) pmk_query_destroy(c->remote);
        c->remote = nullptr;
        if ((rc = pmk_query_create(m, 0, nullptr, &c->remote))) return rc;
        c->remote_model = m;
    }
    c->remote->roff_P = m->P_global;
    if ((rc = query_set_items(c->remote, nrecv, c->d_rx, c->d_rr))) return rc;
    if ((rc = pmk_query_items(c->remote, th))) return rc;
    if (nrecv > 0 && (rc = launch_export_results(c->remote, c->d_ru, c->d_rv, s))) return rc;
    // 3. results back, straight into the requester's sorted item buffers
    PMK_NCCL(g_rccl.GroupStart());
    for (int r = 0; r < W; ++r) {
        if (rcount[(size_t)r] > 0) {
            PMK_NCCL(g_rccl.Send(c->d_ru + rfirst[(size_t)r], (size_t)rcount[(size_t)r], ncclFloat64, r, c->comm, s));
            PMK_NCCL(g_rccl.Send(c->d_rv + rfirst[(size_t)r], (size_t)rcount[(size_t)r], ncclFloat64, r, c->comm, s));
        }
        if (scount[(size_t)r] > 0) {
            PMK_NCCL(g_rccl.Recv(q->d_u + sfirst[(size_t)r], (size_t)scount[(size_t)r], ncclFloat64, r, c->comm, s));
            PMK_NCCL(g_rccl.Recv(q->d_v + sfirst[(size_t)r], (size_t)scount[(size_t)r], ncclFloat64, r, c->comm, s));
        }
    }
    PMK_NCCL(g_rccl.GroupEnd());
    // 4. blend
    return pmk_query_mix(q, weight_th, 0, q->Nq);
}

}  // extern "C"
