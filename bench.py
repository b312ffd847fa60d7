#!/usr/bin/env python3
"""bench.py -- headline benchmark of the per-patch GP hot path on MI355X.

Metric (BASELINE.json): patch-solves/s (+ predict-points/s), 256 patches x 2k points per GPU, fp64,
2-D Spline34 mixGP (config C).  One "step" = one pass of the hot path over one batch of synthetic
input that is already resident in HBM:
    fit step     : kernel-matrix build + Cholesky + weight solves for every patch of this rank
    predict step : partition search + work-item plan, per-(query, region) prediction, all-gather of
                   the per-item (u, v) across ranks (RCCL), mixture -- for every query of the job
`value` = patch-solves/s over all ranks (fit steps timed as the contract says: W warm-up steps, K
timed steps between barrier + synchronize, max over ranks).  predict-points/s is timed the same way
in a second loop and reported beside it.  N GPUs: weak scaling -- every rank owns 256 leaves of one
shared BSP tree (no data-path collective in fit; one all-gather in predict).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--nq NQ] [--no-cpu]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6     # MI355X datasheet fp64 matrix/vector peak (not in the local guide)
FP32_PEAK_TFLOPS = 157.3    # fp32 MFMA peak (MI355X_MICROARCH.md, chip-level parameters)


def oracle_f(X):
    A = np.array([[1.0, 0.4], [0.4, 1.0]]) * 0.1              # examples/mixGP.jl:44-48
    q = np.einsum("ni,ij,nj->n", X, A, X)
    return np.sinc((q / 3.2) ** 2) * (np.linalg.norm(X, axis=1) / 4) ** 3


def chol_flops(n):
    """algorithmic flops of one patch-solve's factor + solves: n^3/3 + 2 n^2"""
    return n ** 3 / 3.0 + 2.0 * n ** 2


def panel_flops(ld, tile=128):
    """flops executed by chol_panel_kernel over one patch: per step k the rows below the diagonal block
    get a GEMM of depth 128 k and a triangular 128-wide solve"""
    nt = ld // tile
    f = 0.0
    for k in range(nt - 1):
        rows = ld - (k + 1) * tile
        f += 2.0 * rows * tile * (tile * k) + rows * tile * tile
        f += tile * tile * (tile * k)        # look-ahead workgroup: symmetric update of the next diagonal tile
    return f


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nq", type=int, default=1 << 20, help="query points per GPU")
    ap.add_argument("--patches", type=int, default=256, help="patches per GPU")
    ap.add_argument("--n", type=int, default=2000, help="points per patch")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--eps", type=float, default=0.0,
                    help="overlap of the training sets (organizetrainingsets); 0.044 gives the ragged 'realistic' "
                         "variant of config C (n ~ 1.8k-2.2k per patch, SURVEY 8(d))")
    ap.add_argument("--config", default="C", choices=["C", "E"],
                    help="C: headline 2-D fp64 256 x 2000 (default).  E: 3-D, 128 x 8192, fp32 (BASELINE config E; "
                         "not the headline: no CPU leg, fp32 MFMA peak)")
    args = ap.parse_args()

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

    import patchmixturekriging_amd as pmk
    from patchmixturekriging_amd import mixture as M
    pmk.set_device(local_rank)
    ctx = pmk.default_context()
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)          # launch on torch's stream: its events then see our kernels

    # ---------------------------------------------------------------- synthetic input (config C, weak-scaled)
    cfgE = args.config == "E"
    if cfgE:
        args.patches, args.n, args.no_cpu = 128, 8192, True
    P, n = args.patches, args.n
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
    Nq = args.nq * world
    Xq = rng.uniform(0, 1, (Nq, 3)) if cfgE else np.stack([rng.uniform(-5, 5, Nq), rng.uniform(-10, 10, Nq)], 1)
    query = pmk.DeviceQuery(model, Xq[rank * args.nq:(rank + 1) * args.nq])   # queries are sharded like the leaves

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
    stage_ms = {"kernel_matrix": [], "cholesky": [], "solve": [], "panel": []}

    def fit_step():
        model.fit(th, sigma2)

    dt_fit = timed(fit_step, args.steps, args.warmup)
    info = model.info()
    assert np.all(info == 0), "a patch was not positive definite"
    # per-stage device times (HIP events on the launch stream) of a few extra steps
    ctx.L.pmk_ctx_enable_timers(ctx.h, 2)
    for _ in range(3):
        fit_step()
        ctx.synchronize()
        for k in stage_ms:
            try:
                stage_ms[k].append(ctx.timer_ms(k))
            except pmk.PmkError:
                pass
    stage = {k: float(np.median(v)) for k, v in stage_ms.items() if v}

    # ---------------------------------------------------------------- predict
    from patchmixturekriging_amd import dist as pdist

    def predict_step():
        # plan of this rank's queries (K5 + sort) -> all-to-all of the (point, region) requests to the leaf owners ->
        # items (K4) -> all-to-all of (u, v) back (RCCL/xGMI) -> mixture (K6)
        return pdist.sharded_predict(query, th, wth, radius, delta, P * world, rank, world)

    ctx.L.pmk_ctx_enable_timers(ctx.h, 0)
    total_items = predict_step()
    dt_pred = timed(predict_step, args.steps, max(1, args.warmup - 1))
    ctx.L.pmk_ctx_enable_timers(ctx.h, 1)
    predict_step()
    ctx.synchronize()
    pstage = {}
    for k in ("plan", "items", "mix"):
        try:
            pstage[k] = ctx.timer_ms(k)
        except pmk.PmkError:
            pass
    Yq, Vq = query.fetch()
    assert np.all(np.isfinite(Yq)) and np.all(Vq >= 1e-12)

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---------------------------------------------------------------- roofline of the dominant kernel
    nmax = max(sizes)
    ld = ((nmax + 127) // 128) * 128
    nt = ld // 128
    roof = None
    if "panel" in stage:
        flops = sum(panel_flops(((s + 127) // 128) * 128) for s in sizes[lo:hi])
        roof = {"bound": "mfma", "kernel": "chol_panel_kernel", "achieved": flops / (stage["panel"] * 1e-3) / 1e12,
                "peak": FP32_PEAK_TFLOPS if cfgE else FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "traffic": None,
                "launches_per_step": nt - 1, "avg_launch_ms": stage["panel"] / (nt - 1),
                "alg_flops_per_launch": flops / (nt - 1)}
    elif "cholesky" in stage:
        flops = sum(chol_flops(s) for s in sizes[lo:hi])
        roof = {"bound": "mfma", "kernel": "chol_diag_kernel+chol_panel_kernel", "achieved": flops / (stage["cholesky"] * 1e-3) / 1e12,
                "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "traffic": None}
    if roof:
        roof["frac"] = roof["achieved"] / roof["peak"]
        # HBM bytes per launch of the dominant kernel from the committed PMC passes (FETCH_SIZE x2 per the gfx950
        # correction + WRITE_SIZE, separate rocprofv3 --pmc runs of this same command; see the file's note)
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            if "panel" in roof["kernel"] and (P, n) == (256, 2000) and args.eps == 0:
                roof["traffic"] = pmc["chol_panel_kernel"]["hbm_bytes_per_launch"]
                roof["traffic_source"] = "profiles/r01_pmc_traffic.json"
        except Exception:
            pass

    # ---------------------------------------------------------------- CPU baseline (oracle = port), rank 0
    cpu = None
    if not args.no_cpu and world == 1:          # the CPU leg runs on rank 0 of the single-GPU run only
        from concurrent.futures import ThreadPoolExecutor
        from oracle import oracle as O
        cores = min(16, os.cpu_count() or 1)
        oth, owth = O.kernel(O.SPLINE34, a), O.kernel(O.SPLINE34, 1 / radius)
        sample = list(range(lo, lo + cores))
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
               "strong_lapack_patch_solves_per_s": strong,
               "sample": "%d of %d patches (n=%d) by the C oracle (kernel loop + LU + Cholesky), one patch per thread"
                         % (len(sample), P, n),
               "predict_points_per_s": len(qs) / t_cpu_pred if len(qs) else None,
               "predict_sample": "%d queries inside the sampled leaves, oracle querymixtureGP! on %d threads" % (len(qs), cores),
               "parity_vs_gpu": {"max_rel_dY": err_y, "max_dV_over_tol": err_v}}

    ms = dt_fit / args.steps * 1e3
    out = {
        "metric": "patch-solves/sec + predict-points/sec, 256 patches x 2k pts" if not cfgE else
                  "patch-solves/sec + predict-points/sec, 128 patches x 8k pts (config E)",
        "value": P * world * args.steps / dt_fit,
        "unit": "patch-solves/s",
        "predict_points_per_s": Nq * args.steps / dt_pred,
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms, "predict_ms_per_step": dt_pred / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype, "data": "synthetic",
        "config": {"workload": ("3-D Spline34(8), %d BSP patches x %d points per GPU, sigma2=1e-3, fp32 (BASELINE config E)"
                                if cfgE else
                                "mixGP 2-D Spline34(1/15), %d BSP patches x %d points per GPU, sigma2=1e-5 (BASELINE config C)")
                               % (P, n),
                   "patches_per_gpu": P, "points_per_patch": n, "patch_sizes_minmax": [min(sizes), max(sizes)],
                   "queries_per_gpu": args.nq, "radius": radius, "items_per_query": total_items / args.nq,
                   "levels": levels, "parallelism": "leaves and queries sharded (%d leaves per GPU); all-to-all of requests and of (u,v)" % P,
                   "bsp_build_s": t_bsp},
        "stage_ms": {**stage, **{"predict_" + k: v for k, v in pstage.items()}},
        "roofline": roof,
        "cpu_baseline": cpu,
    }
    print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
