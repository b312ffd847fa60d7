"""host vs device BSP build time (setuppartition): python tools/bsp_build_time.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import patchmixturekriging_amd as pmk
from patchmixturekriging_amd import _lib

L = _lib.lib()
ctx = pmk.default_context()
for N, levels in ((512000, 9), (2048000, 11), (1048576, 8)):
    D = 3 if levels == 8 else 2
    rng = np.random.Generator(np.random.PCG64(25))
    X = np.ascontiguousarray(rng.uniform(-5, 5, (N, D)))
    res = {}
    for name in ("host", "device", "device"):
        h = C.c_void_p()
        t = time.perf_counter()
        if name == "host":
            rc = L.pmk_bsp_build(D, N, X.ctypes.data_as(C.POINTER(C.c_double)), levels, 1, C.byref(h))
        else:
            rc = L.pmk_bsp_build_device(ctx.h, D, N, X.ctypes.data, levels, 1, C.byref(h))
        res[name] = time.perf_counter() - t
        assert rc == 0
        L.pmk_bsp_destroy(h)
    print("N=%d D=%d levels=%d: host %.1f ms, device %.1f ms (second call, includes the %d MB upload)"
          % (N, D, levels, res["host"] * 1e3, res["device"] * 1e3, N * D * 8 >> 20), flush=True)
