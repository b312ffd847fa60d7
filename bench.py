#!/usr/bin/env python3
"""bench.py -- headline benchmark of the per-patch GP hot path on MI355X.

Metric (BASELINE.json): patch-solves/s (+ predict-points/s), 256 patches x 2k points per GPU, fp64,
2-D Spline34 mixGP (config C).  One "step" = one pass of the hot path over one batch of synthetic
input that is already resident in HBM:
    fit step     : kernel-matrix build + Cholesky + weight solves for every patch of this rank
    predict step : partition search + work-item plan, requests to the leaf owners, per-(query, region)
                   prediction, (u, v) back (RCCL inside libpmk_hip.so), mixture -- for every query of the job
`value` = patch-solves/s over all ranks (fit steps timed as the contract says: W warm-up steps, K
timed steps between barrier + synchronize, max over ranks).  predict-points/s is timed the same way
in a second loop and reported beside it.  N GPUs: weak scaling -- every rank owns `patches` leaves of one
shared BSP tree and `nq` queries (no data-path collective in fit; one request/response exchange in predict).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C|D|E] [--eps E] [--nq NQ] [--no-cpu]

Configs (BASELINE.json): C = headline, 256 x 2000 per GPU.  D = 1024 x 2000 over 8 GPUs: 128 leaves and 524 288 queries
per GPU (run with --gpus 8; `--config D --patches 1024 --nq 4194304` is the whole workload on one GPU).
E = 3-D, 128 x 8192, fp32, checked against the fp64 device model in the same run.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NOMINAL_GHZ = 2.4           # the clock the datasheet peaks are quoted at
FP64_PEAK_TFLOPS = 78.6     # MI355X datasheet fp64 matrix/vector peak (not in the local guide; measured 76.5-77.8)
FP32_PEAK_TFLOPS = 157.3    # fp32 MFMA peak (MI355X_MICROARCH.md, chip-level parameters)
HBM_PEAK_TBS = 8.0          # spec; 6.29 TB/s is what a float4 copy achieves (MI355X_MICROARCH.md)
HBM_ACHIEVABLE_TBS = 6.29


def oracle_f(X):
    A = np.array([[1.0, 0.4], [0.4, 1.0]]) * 0.1              # examples/mixGP.jl:44-48
    q = np.einsum("ni,ij,nj->n", X, A, X)
    return np.sinc((q / 3.2) ** 2) * (np.linalg.norm(X, axis=1) / 4) ** 3


def chol_flops(n):
    """algorithmic flops of one patch-solve's factor + solves: n^3/3 + 2 n^2"""
    return n ** 3 / 3.0 + 2.0 * n ** 2


def executed_step_flops(n, tile=128):
    """flops the step launches execute for one patch: full-rectangle GEMMs on the padded tiles, block substitutions,
    the look-ahead of every diagonal tile"""
    nt = (n + tile - 1) // tile
    f = 0.0
    for k in range(nt - 1):
        rows = (nt - k - 1) * tile
        f += 2.0 * rows * tile * (tile * k) + rows * tile * tile
        f += tile * tile * (tile * (k + 1))
    return f


def kernel_source_hash():
    """identity of the device code this run executes: sha256 over the kernel sources of libpmk_hip.so (what a PMC summary
    must have been taken on for its byte counts to mean anything for this run)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "patchmixturekriging_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


PMC_SUMMARY = os.path.join("profiles", "r03_pmc_summary.json")


def load_pmc_summary():
    """(summary, source) -- the summary only if it was collected on this build of the kernels, else (None, why)"""
    try:
        pmc = json.load(open(os.path.join(ROOT, PMC_SUMMARY)))
    except Exception:
        return None, "no %s" % PMC_SUMMARY
    have, want = pmc.get("kernel_source_hash"), kernel_source_hash()
    if have != want:
        return None, "%s is stale (kernel sources %s, summary taken on %s): traffic dropped" % (PMC_SUMMARY, want, have)
    return pmc, PMC_SUMMARY


def host_cores():
    """threads the CPU leg may use: the affinity mask, capped by the cgroup CPU quota of the box (a GPU box may show every
    core of the host in its mask while the container is entitled to a share of them)"""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().split()
        if txt[0] != "max":
            quota = float(txt[0]) / float(txt[1])
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except Exception:
            pass
    cores = aff if quota is None else max(1, min(aff, int(np.ceil(quota))))
    return cores, aff, quota


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nq", type=int, default=None, help="query points per GPU (default: 2^20; config D: 524288)")
    ap.add_argument("--patches", type=int, default=None, help="patches per GPU (default 256; D: 128; E: 128)")
    ap.add_argument("--n", type=int, default=None, help="points per patch (default 2000; E: 8192)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--eps", type=float, default=0.0,
                    help="overlap of the training sets (organizetrainingsets); 0.044 gives the ragged 'realistic' "
                         "variant of config C (n ~ 1.8k-2.2k per patch, SURVEY 8(d))")
    ap.add_argument("--config", default="C", choices=["C", "D", "E"])
    ap.add_argument("--exchange", default=os.environ.get("PMK_BENCH_EXCHANGE", "abi"),
                    choices=["abi", "torch", "allgather", "allgather-torch"],
                    help="N > 1: 'abi' = pmk_query_predict_sharded (requests to the leaf owners and (u, v) back, the library's "
                         "own RCCL communicator); 'torch' = the same step driven from Python over torch.distributed "
                         "(patchmixturekriging_amd/dist.py); 'allgather' = BASELINE.json's literal form, "
                         "pmk_query_predict_allgather: REPLICATED queries, every rank plans all of them, one ncclAllGather of "
                         "padded (u, v) slices; 'allgather-torch' = that form over torch.distributed")
    ap.add_argument("--shard-queries", default="index", choices=["index", "home"],
                    help="which queries a rank holds (request/response exchanges): 'index' = an equal slice of the query "
                         "array; 'home' = the queries whose home leaf it owns (routed once, before the timed region: only "
                         "items that cross a subtree boundary then travel)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end_to_end block (reference-named API wall times)")
    args = ap.parse_args()

    if "RANK" not in os.environ and args.gpus > 1:
        # launched plainly (python bench.py --gpus N): start N fresh ranks and relay rank 0's line.  This process has not
        # imported torch or touched a GPU, and it never replaces itself: the ranks are children.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus))
        procs = []
        for r_ in range(args.gpus):            # what torch.distributed.run would set, one child per rank
            e = dict(env, RANK=str(r_), LOCAL_RANK=str(r_))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                          stdout=subprocess.PIPE if r_ == 0 else subprocess.DEVNULL, text=True))
        import threading
        buf = []
        rd = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()))
        rd.start()
        while any(p_.poll() is None for p_ in procs):
            if any(p_.poll() not in (None, 0) for p_ in procs):     # a rank died: its peers would wait in a collective for ever
                for p_ in procs:
                    if p_.poll() is None:
                        p_.kill()
                break
            time.sleep(0.2)
        rcs = [p_.wait() for p_ in procs]
        rd.join()
        out0 = buf[0] if buf else ""
        lines = [ln for ln in out0.splitlines() if ln.startswith("{") and '"metric"' in ln]
        for ln in out0.splitlines():
            if ln not in lines:
                print(ln, file=sys.stderr)
        if lines:
            print(lines[-1])
        rc = next((c for c in rcs if c), 0)
        sys.exit(rc if rc else (0 if lines else 1))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    import torch.distributed as dist
    # PMK_BENCH_BACKEND=gloo + PMK_BENCH_SHARE_GPU=1: rehearse the N>1 path on a one-GPU box (every rank on cuda:0)
    backend = os.environ.get("PMK_BENCH_BACKEND", "nccl")
    if os.environ.get("PMK_BENCH_SHARE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
            # two ranks on one device: RCCL refuses duplicate GPUs, so the library's communicator is out
            args.exchange = {"abi": "torch", "allgather": "allgather-torch"}.get(args.exchange, args.exchange)

    import patchmixturekriging_amd as pmk
    from patchmixturekriging_amd import dist as pdist
    pmk.set_device(local_rank)
    ctx = pmk.default_context()
    pdist.use_torch_stream(ctx)                 # launch on torch's stream: its events and collectives order with our kernels

    # ---------------------------------------------------------------- synthetic input, weak-scaled
    cfg = args.config
    defaults = {"C": (256, 2000, 1 << 20), "D": (128, 2000, 1 << 19), "E": (128, 8192, 1 << 20)}[cfg]
    P = args.patches or defaults[0]
    n = args.n or defaults[1]
    nq = args.nq or defaults[2]
    cfgE = cfg == "E"
    levels_local = int(round(np.log2(P))) + 1
    assert 2 ** (levels_local - 1) == P, "--patches must be a power of two"
    levels = levels_local + int(round(np.log2(world)))
    assert 2 ** (levels - 1) == P * world, "--gpus must be a power of two"
    N = P * n * world
    a, sigma2, delta = 1 / 15, 1e-5, 1e-5
    rng = np.random.Generator(np.random.PCG64(25))
    if cfgE:
        # unit cube, compact kernel (support ~0.6 patch widths) and a noise level that keep cond(U) * eps32 << 1
        a, sigma2, delta = 8.0, 1e-3, 1e-6
        X = rng.uniform(0, 1, (N, 3))
        y = np.sin(3 * X[:, 0]) * np.cos(2 * X[:, 1]) + X[:, 2] ** 2
    else:
        X = np.stack([rng.uniform(-5, 5, N), rng.uniform(-10, 10, N)], 1)
        y = oracle_f(X)
    t0 = time.time()
    root, X_parts, X_parts_inds = pmk.setuppartition(X, levels, device=True)   # GPU build, bit-identical to the host's
    t_bsp = time.time() - t0
    if args.eps > 0:      # overlapping training sets: ragged patch sizes
        X_parts, X_parts_inds, _, _ = pmk.organizetrainingsets(root, levels, X, args.eps)
    sizes = [len(p) for p in X_parts]
    # radius = 0.1 x patch width: patch area = 200 / (P world)
    radius = 0.1 * (1.0 / (P * world)) ** (1 / 3) if cfgE else 0.1 * np.sqrt(200.0 / (P * world))
    th, wth = pmk.Spline34KernelType(a), pmk.Spline34KernelType(1 / radius)
    lo, hi = rank * P, (rank + 1) * P
    dtype = "f32" if cfgE else "f64"
    model = pmk.DeviceModel(X_parts[lo:hi], [y[i] for i in X_parts_inds[lo:hi]], dtype=dtype)
    model.set_bsp(root, lo)
    Nq = nq * world
    Xq = rng.uniform(0, 1, (Nq, 3)) if cfgE else np.stack([rng.uniform(-5, 5, Nq), rng.uniform(-10, 10, Nq)], 1)
    replicated = world > 1 and args.exchange.startswith("allgather")
    if replicated:
        Xq_mine = Xq                                 # the all-gather form: every rank holds (and plans) every query
    elif world > 1 and args.shard_queries == "home":
        # route once, outside the timed region: a rank holds the queries whose home leaf it owns
        tmp = pmk.DeviceQuery(model, Xq); tmp.plan(radius, delta)
        home = tmp.debug()["home"]
        del tmp
        Xq_mine = Xq[(home >= lo) & (home < hi)]
    else:
        Xq_mine = Xq[rank * nq:(rank + 1) * nq]      # queries are sharded like the leaves
    query = pmk.DeviceQuery(model, Xq_mine)

    comm, exchange = None, "none"
    if world > 1:
        exchange = args.exchange
        if exchange in ("abi", "allgather"):
            try:
                idt = torch.zeros(pmk.context.COMM_ID_BYTES, dtype=torch.uint8, device="cuda")
                if rank == 0:
                    idt.copy_(torch.frombuffer(bytearray(pmk.comm_unique_id()), dtype=torch.uint8))
                dist.broadcast(idt, 0)
                comm = pmk.Comm(ctx, rank, world, bytes(idt.cpu().numpy().tobytes()))
            except Exception as e:                                  # say so loudly; the step itself is the same
                print("bench.py: the library's RCCL communicator could not be created (%s); using torch.distributed" % e,
                      file=sys.stderr)
                exchange = {"abi": "torch", "allgather": "allgather-torch"}[exchange]
            flag = torch.tensor([1 if exchange.endswith("torch") else 0], device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)              # all ranks must take the same path
            if int(flag.item()) and not exchange.endswith("torch"):
                comm.close()
                comm, exchange = None, {"abi": "torch", "allgather": "allgather-torch"}[exchange]

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        sync()
        t = time.perf_counter()
        for _ in range(steps):
            fn()
        sync()
        dt = time.perf_counter() - t
        if world > 1:
            tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    # ---------------------------------------------------------------- fit
    def fit_step():
        model.fit(th, sigma2)

    dt_fit = timed(fit_step, args.steps, args.warmup)
    info = model.info()
    assert np.all(info == 0), "a patch was not positive definite"
    # per-stage device times (HIP events on the launch stream) of a few extra steps
    nt_max = (max(sizes[lo:hi]) + 127) // 128
    stage_ms = {"kernel_matrix": [], "cholesky": [], "solve": [], "panel": []}
    step_us = []
    ctx.L.pmk_ctx_enable_timers(ctx.h, 2)
    for _ in range(3):
        fit_step()
        ctx.synchronize()
        for k in stage_ms:
            try:
                stage_ms[k].append(ctx.timer_ms(k))
            except pmk.PmkError:
                pass
        us = []
        for i in range(nt_max):                  # however many launches the factorisation made (the queue form: one)
            try:
                us.append(ctx.timer_ms("step:%d" % i) * 1e3)
            except pmk.PmkError:
                break
        if us:
            step_us.append(us)
    stage = {k: float(np.median(v)) for k, v in stage_ms.items() if v}
    if "panel" in stage:
        stage["chol_steps"] = stage.pop("panel")        # the nt - 1 chol_step_kernel launches of a fit

    # ---------------------------------------------------------------- predict
    def predict_step():
        # plan of this rank's queries (K5 + sort) -> requests to the leaf owners -> items (K4) -> (u, v) back (RCCL over
        # xGMI) -> mixture (K6)
        if exchange == "allgather":
            return query.predict_allgather(comm, th, wth, radius, delta)
        if exchange == "allgather-torch":
            return pdist.allgather_predict(query, th, wth, radius, delta, P * world, rank, world)[0]
        if comm is not None:
            return query.predict_sharded(comm, th, wth, radius, delta)
        return pdist.sharded_predict(query, th, wth, radius, delta, P * world, rank, world)

    ctx.L.pmk_ctx_enable_timers(ctx.h, 0)
    total_items = predict_step()
    dt_pred = timed(predict_step, args.steps, max(1, args.warmup - 1))
    xbytes = None
    if world > 1:
        if comm is not None:
            sent, recv = comm.last_bytes()
        elif exchange == "allgather-torch":
            recv = pdist.allgather_predict(query, th, wth, radius, delta, P * world, rank, world)[1]; sent = recv
        else:
            sent = recv = None
        if sent is not None:
            tb = torch.tensor([sent, recv, total_items, len(Xq_mine)], device="cuda", dtype=torch.int64)
            tl = [torch.zeros_like(tb) for _ in range(world)]
            dist.all_gather(tl, tb)
            xbytes = {"sent_per_rank": [int(t[0]) for t in tl], "received_per_rank": [int(t[1]) for t in tl],
                      "items_per_rank": [int(t[2]) for t in tl], "queries_per_rank": [int(t[3]) for t in tl]}
    ctx.L.pmk_ctx_enable_timers(ctx.h, 1)
    predict_step()
    ctx.synchronize()
    pstage = {}
    for k in ("plan", "items", "mix"):
        try:
            pstage[k] = ctx.timer_ms(k)
        except pmk.PmkError:
            pass
    ctx.L.pmk_ctx_enable_timers(ctx.h, 0)
    Yq, Vq = query.fetch()
    assert np.all(np.isfinite(Yq)) and np.all(Vq >= 1e-12)

    # ---------------------------------------------------------------- config E: the fp32 result against the fp64 device model
    parity32 = None
    if cfgE and world == 1:
        m64 = pmk.DeviceModel(X_parts[lo:hi], [y[i] for i in X_parts_inds[lo:hi]])
        m64.fit(th, sigma2)
        assert np.all(m64.info() == 0)
        m64.set_bsp(root, lo)
        ns = min(nq, 1 << 16)
        q64 = pmk.DeviceQuery(m64, Xq[:ns]); q64.plan(radius, delta); q64.items(th); q64.mix(wth)
        Y64, V64 = q64.fetch()
        q32 = pmk.DeviceQuery(model, Xq[:ns]); q32.plan(radius, delta); q32.items(th); q32.mix(wth)
        Y32, V32 = q32.fetch()
        d64, d32 = q64.debug(), q32.debug()
        ids = all(np.array_equal(d64[k], d32[k]) for k in ("home", "item_offsets", "item_region", "item_t"))
        c64, c32 = m64.get(0, 0), model.get(0, 0)
        parity32 = {"queries": ns, "ids_and_t_identical": bool(ids),
                    "max_rel_dY": float(np.max(np.abs(Y32 - Y64) / np.maximum(1, np.abs(Y64)))),
                    "max_abs_dV": float(np.max(np.abs(V32 - V64))), "V_range": [float(V64.min()), float(V64.max())],
                    "rel_dc_patch0": float(np.linalg.norm(c32 - c64) / np.linalg.norm(c64)),
                    "bounds": "dY <= 1e-4, dV <= 5e-5 + 2e-3 V (tests/test_gpu_parity.py, eps32-scaled)"}
        assert ids and parity32["max_rel_dY"] <= 1e-4 and np.all(np.abs(V32 - V64) <= 5e-5 + 2e-3 * V64), parity32
        del q64, m64

    if rank != 0:
        if world > 1:
            dist.barrier()
            if comm is not None:
                comm.close()
            dist.destroy_process_group()
        return

    # ---------------------------------------------------------------- rooflines (algorithmic work / device time)
    mine = sizes[lo:hi]
    peak = FP32_PEAK_TFLOPS if cfgE else FP64_PEAK_TFLOPS
    esz = 4 if cfgE else 8
    D = X.shape[1]
    alg = sum(chol_flops(s) for s in mine)
    roof = None
    if "chol_steps" in stage and nt_max > 1:
        launches = len(step_us[0]) if step_us else nt_max - 1
        t_steps = stage["chol_steps"] * 1e-3
        roof = {"bound": "mfma", "kernel": "chol_step_kernel",
                "achieved": alg / t_steps / 1e12, "peak": peak, "unit": "TFLOP/s",
                "frac": alg / t_steps / 1e12 / peak,
                "traffic": None,
                "launches_per_step": launches, "avg_launch_ms": stage["chol_steps"] / launches,
                "alg_flops_per_launch": alg / launches,
                "note": "algorithmic flops = sum over patches of n^3/3 + 2 n^2 (real n), / the device time of the nt-1 step "
                        "launches (HIP events on the launch stream)",
                "executed_flops_frac": sum(executed_step_flops(s) for s in mine) / t_steps / 1e12 / peak,
                "step_launch_us": [round(float(v), 1) for v in np.median(np.array(step_us), axis=0)] if step_us else None}
        ghz = ctx.shader_clock(0)
        if ghz > 0:     # the peak is the datasheet's (2.4 GHz); the chip clocks lower under this kernel
            roof["shader_clock_ghz"] = ghz
            roof["frac_of_peak_at_that_clock"] = roof["frac"] * NOMINAL_GHZ / ghz
            roof["executed_flops_frac_at_that_clock"] = roof["executed_flops_frac"] * NOMINAL_GHZ / ghz
        # HBM bytes per launch from the committed PMC passes of this same command (FETCH_SIZE x2 + WRITE_SIZE) -- only if
        # they were taken on THIS build of the kernels (the summary carries the hash of the kernel sources)
        pmc, pmc_src = load_pmc_summary()
        if pmc is not None and cfg == "C" and (P, n) == (256, 2000) and args.eps == 0 and "chol_step_kernel" in pmc:
            roof["traffic"] = pmc["chol_step_kernel"].get("hbm_bytes_per_dispatch")
            roof["traffic_source"] = pmc_src
        elif pmc is None:
            roof["traffic_source"] = pmc_src
    ms = dt_fit / args.steps * 1e3
    roof_fit = {"bound": "mfma", "achieved": alg / (ms * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s",
                "frac": alg / (ms * 1e-3) / 1e12 / peak, "note": "algorithmic flops / ms_per_step (K1 + factor + solves)"}
    roof_k1 = None
    if "kernel_matrix" in stage:
        k1_bytes = sum(esz * (s * (s + 1) / 2 + s * D) for s in mine)
        roof_k1 = {"bound": "hbm", "kernel": "kmat_slab_kernel", "achieved": k1_bytes / (stage["kernel_matrix"] * 1e-3) / 1e9,
                   "peak": HBM_PEAK_TBS * 1e3, "unit": "GB/s", "frac": k1_bytes / (stage["kernel_matrix"] * 1e-3) / 1e12 / HBM_PEAK_TBS,
                   "frac_of_achievable_6.29": k1_bytes / (stage["kernel_matrix"] * 1e-3) / 1e12 / HBM_ACHIEVABLE_TBS,
                   "note": "algorithmic bytes = s (n(n+1)/2 + n D) per patch: the lower triangle is what is materialised"}
    roof_pred = None
    if "items" in pstage and world == 1:
        dbg = query.debug()
        regs = dbg["item_region"]
        nn = np.array(sizes, dtype=np.float64)[regs]
        pf = float(np.sum(nn * nn + 4 * nn))
        roof_pred = {"bound": "mfma", "kernel": "predict_strip_kernel", "achieved": pf / (pstage["items"] * 1e-3) / 1e12,
                     "peak": peak, "unit": "TFLOP/s", "frac": pf / (pstage["items"] * 1e-3) / 1e12 / peak,
                     "note": "n^2 + 4 n flops per (query, region) pair / device time of the strip kernel"}
        ghz = ctx.shader_clock(1)
        if ghz > 0:
            roof_pred["shader_clock_ghz"] = ghz
            roof_pred["frac_of_peak_at_that_clock"] = roof_pred["frac"] * NOMINAL_GHZ / ghz
        pmc, pmc_src = load_pmc_summary()
        if pmc is not None and cfg == "C" and (P, n, nq) == (256, 2000, 1 << 20) and args.eps == 0 and "predict_strip_kernel" in pmc:
            roof_pred["traffic"] = pmc["predict_strip_kernel"].get("hbm_bytes_per_dispatch")
            roof_pred["traffic_source"] = pmc_src

    # ---------------------------------------------------------------- end to end: the reference's own timing points
    # examples/mixGP.jl:154 `@time fitmixtureGP!` and :176 `@time querymixtureGP!`, through the reference-named API with host
    # buffers on both sides (model creation, H2D of X / y / Xq, D2H of c, Yq, Vq included): wall clock, median of >= 10
    e2e = None
    if world == 1 and not args.no_e2e and not cfgE:
        reps = 10
        hps = pmk.fetchhyperplanes(root)
        Y_set = [y[i] for i in X_parts_inds]
        eta = pmk.MixtureGPType(X_parts, hps)
        t_fit, t_q = [], []
        for _ in range(reps + 1):
            t = time.perf_counter()
            pmk.fitmixtureGP_(eta, Y_set, th, sigma2)
            t_fit.append(time.perf_counter() - t)
        nq_e = min(Nq, 1 << 20)
        Yo, Vo = np.empty(0), np.empty(0)
        for _ in range(reps + 1):
            t = time.perf_counter()
            Yo, Vo = pmk.querymixtureGP_(Yo, Vo, Xq[:nq_e], eta, root, levels, radius, delta, th, sigma2, wth,
                                         pmk.MixtureGPDebugType(1.0)) or (Yo, Vo)
            t_q.append(time.perf_counter() - t)
        e2e = {"fitmixtureGP_ms": float(np.median(t_fit[1:]) * 1e3), "querymixtureGP_ms": float(np.median(t_q[1:]) * 1e3),
               "patch_solves_per_s": P / float(np.median(t_fit[1:])), "predict_points_per_s": nq_e / float(np.median(t_q[1:])),
               "queries": nq_e, "repetitions": reps,
               "what": "host wall clock of the reference-named calls (mixture.py mirrors of mixtureGP.jl:70-118, 159-294): "
                       "device model created and X, y uploaded inside fitmixtureGP_, c downloaded; Xq uploaded and Yq, Vq "
                       "downloaded inside querymixtureGP_; median of %d after one warm-up" % reps}
        del eta

    # ---------------------------------------------------------------- CPU baseline (oracle = port), rank 0, N = 1
    cpu = None
    if not args.no_cpu and world == 1 and not cfgE:
        from concurrent.futures import ThreadPoolExecutor
        from oracle import oracle as O
        cores, aff_cores, quota = host_cores()
        cores = min(cores, int(os.environ.get("PMK_BENCH_CPU_THREADS", cores)))
        oth, owth = O.kernel(O.SPLINE34, a), O.kernel(O.SPLINE34, 1 / radius)
        nsamp = min(cores, P)
        sample = list(range(lo, lo + nsamp))
        # faithful flavour: what the Julia code does algorithmically, ONE thread -- scalar kernel
        # loop, LU solve for the weights, a second factorisation (Cholesky) for the variance; one patch
        t = time.perf_counter()
        O.fit_patch(oth, X_parts[sample[0]], y[X_parts_inds[sample[0]]], sigma2)
        t_one = time.perf_counter() - t
        t = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:         # ctypes releases the GIL: one patch per core
            fits = list(ex.map(lambda r: O.fit_patch(oth, X_parts[r], y[X_parts_inds[r]], sigma2), sample))
        t_cpu_fit = time.perf_counter() - t
        # predict: queries whose home and neighbours all fall in the sampled (contiguous) leaves
        ob = O.BSP(X, levels)
        dbg = query.debug()
        offs, regs = dbg["item_offsets"], dbg["item_region"]
        inside = np.zeros(P * world, bool)
        inside[sample] = True
        ok = np.add.reduceat(~inside[regs], offs[:-1]) == 0
        qs = np.nonzero(ok)[0][:4000]
        Xs = [X_parts[r] if inside[r] else X_parts[r][:1] for r in range(P * world)]
        cs = [fits[r - lo]["c_lu"] if inside[r] else np.zeros(1) for r in range(P * world)]
        Ls = [fits[r - lo]["L"] if inside[r] else np.ones((1, 1)) for r in range(P * world)]
        t = time.perf_counter()
        oY, oV = O.query_mixture(ob, oth, owth, Xs, cs, Ls, Xq[qs], radius, delta, nthreads=cores)
        t_cpu_pred = time.perf_counter() - t
        t = time.perf_counter()
        nq1 = min(len(qs), 200)
        if nq1:
            O.query_mixture(ob, oth, owth, Xs, cs, Ls, Xq[qs[:nq1]], radius, delta, nthreads=1)
        t_pred_one = time.perf_counter() - t
        err_y = float(np.max(np.abs(Yq[qs] - oY) / np.maximum(1, np.abs(oY)))) if len(qs) else 0.0
        err_v = float(np.max(np.abs(Vq[qs] - oV) / (1e-9 + 1e-5 * oV))) if len(qs) else 0.0
        # "strong" flavour: same kernel loop, factorisations by LAPACK (scipy = OpenBLAS, the
        # library family Julia's `\\` and `cholesky` reach); one patch per host thread, BLAS kept single-threaded so
        # the pool does not oversubscribe the box's CPU share
        strong = None
        try:
            import scipy.linalg as sla
            from threadpoolctl import threadpool_limits

            def strong_fit(r):
                K = O.kernel_matrix(oth, X_parts[r])
                K[np.diag_indices_from(K)] += sigma2
                c_ = sla.lu_solve(sla.lu_factor(K, check_finite=False), y[X_parts_inds[r]], check_finite=False)
                return c_, sla.cholesky(K, lower=True, check_finite=False)

            with threadpool_limits(limits=1):
                strong_fit(sample[0])                               # warm the BLAS pool outside the timed region
                t = time.perf_counter()
                with ThreadPoolExecutor(cores) as ex:
                    list(ex.map(strong_fit, sample))
                strong = len(sample) / (time.perf_counter() - t)
        except Exception:
            pass
        cpu = {"value": len(sample) / t_cpu_fit, "unit": "patch-solves/s", "cores": cores, "kind": "port",
               "cpu_count": os.cpu_count(), "affinity_cores": aff_cores, "cgroup_cpu_quota": quota,
               "faithful_1thread": {"patch_solves_per_s": 1.0 / t_one,
                                    "predict_points_per_s": nq1 / t_pred_one if nq1 else None,
                                    "what": "C oracle on ONE thread: scalar kernel loop, LU for c, Cholesky for L; per-query TRSV"},
               "strong_lapack_patch_solves_per_s": strong,
               "sample": "%d of %d patches (n=%d) by the C oracle (kernel loop + LU + Cholesky), one patch per thread on the "
                         "%d cores this box is entitled to (affinity mask capped by the cgroup quota)" % (len(sample), P, n, cores),
               "predict_points_per_s": len(qs) / t_cpu_pred if len(qs) else None,
               "predict_sample": "%d queries inside the sampled leaves, oracle querymixtureGP! on %d threads" % (len(qs), cores),
               "parity_vs_gpu": {"max_rel_dY": err_y, "max_dV_over_tol": err_v}}

    metric = {"C": "patch-solves/sec + predict-points/sec, 256 patches x 2k pts",
              "D": "patch-solves/sec + predict-points/sec, 1024 patches x 2k pts over 8 GPUs (config D: 128 leaves per GPU)",
              "E": "patch-solves/sec + predict-points/sec, 128 patches x 8k pts (config E)"}[cfg]
    work = {"C": "mixGP 2-D Spline34(1/15), %d BSP patches x %d points per GPU, sigma2=1e-5 (BASELINE config C)",
            "D": "mixGP 2-D Spline34(1/15), %d BSP patches x %d points per GPU, sigma2=1e-5 (BASELINE config D shape)",
            "E": "3-D Spline34(8), %d BSP patches x %d points per GPU, sigma2=1e-3, fp32 (BASELINE config E)"}[cfg] % (P, n)
    out = {
        "metric": metric,
        "value": P * world * args.steps / dt_fit,
        "unit": "patch-solves/s",
        "predict_points_per_s": Nq * args.steps / dt_pred,
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms, "predict_ms_per_step": dt_pred / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype, "data": "synthetic",
        "config": {"workload": work,
                   "patches_per_gpu": P, "points_per_patch": n, "patch_sizes_minmax": [min(sizes), max(sizes)],
                   "eps": args.eps, "queries_per_gpu": nq, "radius": radius, "items_per_query": total_items / nq,
                   "levels": levels,
                   "parallelism": ("leaves sharded (%d per GPU), queries REPLICATED; one RCCL all-gather of padded (u,v) slices" % P)
                                  if replicated else
                                  ("leaves and queries sharded (%d leaves per GPU, queries by %s); requests to the leaf owners and "
                                   "(u,v) back by grouped RCCL send/recv" % (P, args.shard_queries)),
                   "exchange": exchange, "shard_queries": "replicated" if replicated else args.shard_queries,
                   "bsp_build_s": t_bsp},
        "stage_ms": {**stage, **{"predict_" + k: v for k, v in pstage.items()}},
        "roofline": roof,
        "roofline_fit_step": roof_fit,
        "roofline_kernel_matrix": roof_k1,
        "roofline_predict": roof_pred,
        "cpu_baseline": cpu,
        "end_to_end": e2e,
        "kernel_source_hash": kernel_source_hash(),
    }
    if xbytes is not None:
        out["exchange_bytes"] = xbytes
    if parity32 is not None:
        out["parity_vs_fp64"] = parity32
    print(json.dumps(out))
    if world > 1:
        dist.barrier()
        if comm is not None:
            comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
