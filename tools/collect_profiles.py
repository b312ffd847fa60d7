#!/usr/bin/env python3
"""Copy the evidence that tools/refresh_profiles.sh left under gpurun_out/<tag>/ into profiles/ (tracked):
    python tools/collect_profiles.py r02b r02
kernel stats, bench lines, the PMC summary, the command lines, and ONE raw counter CSV (the rows of this library's
kernels from every pass, kernel names shortened, a Pass column added) -- the per-pass CSVs themselves are megabytes."""
import csv
import glob
import os
import re
import shutil
import sys


def main():
    src_tag, dst_tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "gpurun_out", src_tag)
    dst = os.path.join(root, "profiles")
    for a, b in (("kernel_stats.csv", "kernel_stats.csv"), ("kernel_stats_bench.json", "kernel_stats_bench.json"),
                 ("bench_C.json", "bench_C.json"), ("ragged_bench.json", "ragged_bench.json"),
                 ("config_E_fp32_bench.json", "config_E_fp32_bench.json"),
                 ("config_D_one_gpu_bench.json", "config_D_one_gpu_bench.json"),
                 ("config_E_fp32_unfused_bench.json", "config_E_fp32_unfused_bench.json"),
                 ("single_problem.json", "single_problem.json"), ("e2e_breakdown.txt", "e2e_breakdown.txt"),
                 ("pmc/summary.json", "pmc_summary.json"), ("commands.txt", "pmc_commands.txt")):
        if os.path.exists(os.path.join(src, a)):
            shutil.copy(os.path.join(src, a), os.path.join(dst, "%s_%s" % (dst_tag, b)))
    rows, header = [], None
    for d in sorted(glob.glob(os.path.join(src, "pmc", "*", ""))):
        f = glob.glob(os.path.join(d, "*counter_collection.csv"))
        if not f:
            continue
        pas = os.path.basename(os.path.dirname(d))
        with open(f[0]) as fh:
            rd = csv.reader(fh)
            h = next(rd)
            header = header or h + ["Pass"]
            ki = h.index("Kernel_Name")
            for r in rd:
                if "pmk::" not in r[ki]:
                    continue
                m = re.search(r"pmk::(?:f64::|f32::|\(anonymous namespace\)::)?(\w+)", r[ki])
                r[ki] = m.group(1) if m else r[ki][:40]
                rows.append(r + [pas])
    if rows:
        with open(os.path.join(dst, "%s_pmc_counters_raw.csv" % dst_tag), "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(header)
            w.writerows(rows)
    print("collected %d counter rows" % len(rows))


if __name__ == "__main__":
    main()
