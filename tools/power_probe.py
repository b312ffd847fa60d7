#!/usr/bin/env python3
"""Board power and clocks (rocm-smi, sampled from a thread) while fits run back to back, task-queue factorisation
against one launch per block column, same process, same box."""
import json, os, subprocess, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import patchmixturekriging_amd as pmk


def sampler(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "-P", "-c", "-t", "--json"], capture_output=True, text=True, timeout=5)
            d = json.loads(r.stdout)
            c = d.get("card0", {})
            out.append((time.time(), c))
        except Exception as e:
            out.append((time.time(), {"err": str(e)}))


def main():
    P, n, levels = 256, 2000, 9
    rng = np.random.Generator(np.random.PCG64(25))
    X = np.stack([rng.uniform(-5, 5, P * n), rng.uniform(-10, 10, P * n)], 1)
    y = np.sin(X[:, 0]) * np.cos(0.3 * X[:, 1])
    root, Xp, Xi = pmk.setuppartition(X, levels, device=True)
    ctx = pmk.default_context()
    th = pmk.Spline34KernelType(1 / 15)
    secs = float(os.environ.get("PP_SECS", "4"))
    ref = None
    for spec in os.environ.get("PP_MODES", "1:1,0:1,1:1,0:1").split(","):
        mode, segs, frm = (spec.split(":") + ["0"])[:3]
        os.environ["PMK_CHOL_QUEUE"] = mode
        os.environ["PMK_QUEUE_SEGS"] = segs
        os.environ["PMK_QUEUE_FROM"] = frm
        model = pmk.DeviceModel(Xp, [y[i] for i in Xi])
        for _ in range(3):
            model.fit(th, 1e-5)
        ctx.synchronize()
        assert np.all(model.info() == 0)
        cs = np.concatenate([model.get(r, 0) for r in (0, 17, 255)])
        if ref is None:
            ref = cs
        same = bool(np.array_equal(ref, cs))
        stop, out = threading.Event(), []
        th_ = threading.Thread(target=sampler, args=(stop, out)); th_.start()
        t0 = time.time(); nfit = 0
        while time.time() - t0 < secs:
            for _ in range(10):
                model.fit(th, 1e-5)
            ctx.synchronize(); nfit += 10
        dt = time.time() - t0
        stop.set(); th_.join()
        clk = ctx.shader_clock(0) if hasattr(ctx, "shader_clock") else None
        keys = set()
        for _, c in out: keys |= set(c.keys())
        summ = {}
        for kx in sorted(keys):
            vals = []
            for _, c in out:
                v = c.get(kx)
                try: vals.append(float(str(v).strip("()MhzW ").split()[0]))
                except Exception: pass
            if vals: summ[kx] = (round(float(np.mean(vals)), 1), round(float(np.max(vals)), 1))
        print(json.dumps({"queue": mode, "segs": segs, "from": frm, "c_identical_to_first": same, "ms_per_fit": 1e3 * dt / nfit, "probe_ghz": clk, "samples": len(out), "smi_mean_max": summ}))
        del model


if __name__ == "__main__":
    main()
