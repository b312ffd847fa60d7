#!/usr/bin/env python3
"""Reduce the rocprofv3 --pmc CSVs written by tools/pmc_passes.sh: per kernel and counter, the number of
dispatches and the mean counter value per dispatch; FETCH_SIZE / WRITE_SIZE are in KB (FETCH_SIZE is doubled for
the gfx950 under-count of wide streaming reads, MI355X_MICROARCH.md section HBM)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    for key in ("predict_strip_kernel", "chol_step_kernel", "chol_first_kernel", "chol_backsolve_kernel",
                "kmat_slab_kernel", "plan_kernel", "mix_kernel"):
        if key in name:
            return key
    return name[-60:]


def main():
    root = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                c = row["Counter_Name"]
                a = acc[k][c]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    out = {}
    for k, cs in sorted(acc.items()):
        out[k] = {}
        for c, (n, tot) in cs.items():
            out[k][c] = {"dispatches": n, "mean_per_dispatch": tot / n}
        d = out[k]
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["hbm_bytes_per_dispatch"] = (2 * d["FETCH_SIZE"]["mean_per_dispatch"] + d["WRITE_SIZE"]["mean_per_dispatch"]) * 1024
        if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d:
            h, m = d["TCC_HIT_sum"]["mean_per_dispatch"], d["TCC_MISS_sum"]["mean_per_dispatch"]
            d["l2_hit_rate"] = h / max(h + m, 1.0)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
            # MFMA-busy cycles summed over SIMDs / (GPU-active cycles x 1024 SIMDs); GRBM_GUI_ACTIVE sums the 8 XCDs
            d["mfma_busy_frac"] = d["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_dispatch"] / max(
                d["GRBM_GUI_ACTIVE"]["mean_per_dispatch"] / 8.0 * 1024.0, 1.0)
    # the build the counters were taken on: bench.py only quotes these byte counts for the same kernel sources
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_hash
    out["kernel_source_hash"] = kernel_source_hash()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
