#!/bin/bash
# HBM traffic / L2 hit / issue-stall PMC passes of one bench configuration (run on the GPU box through gpurun):
#     tools/pmc_passes.sh <outdir> [bench.py args...]
# one rocprofv3 --pmc pass per counter group (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950), each with
# --kernel-trace only; tools/pmc_reduce.py turns the CSVs into per-kernel, per-launch numbers.
set -e
out=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"; do
    tag=$(echo "$grp" | tr ' ' '+' | cut -c1-60)
    echo "== pass $tag: rocprofv3 --pmc $grp --kernel-trace --output-format csv -- python3 bench.py $*" | tee -a "$out/commands.txt"
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$out/$tag" -o p -- python3 bench.py "$@" > "$out/$tag.log" 2>&1
    rm -f "$out/$tag"/*kernel_trace.csv
done
python3 tools/pmc_reduce.py "$out" > "$out/summary.json"
python3 - "$out/summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if isinstance(v, dict) and "hbm_bytes_per_dispatch" in v:
        print("%-24s HBM %.3f GB/dispatch (fetch x2 %.3f + write %.3f), L2 hit %.3f, mfma_busy %.3f, dispatches %d" % (
            k, v["hbm_bytes_per_dispatch"] / 1e9, 2 * v["FETCH_SIZE"]["mean_per_dispatch"] * 1024 / 1e9,
            v["WRITE_SIZE"]["mean_per_dispatch"] * 1024 / 1e9, v.get("l2_hit_rate", -1), v.get("mfma_busy_frac", -1),
            v["FETCH_SIZE"]["dispatches"]))
        if "SQ_WAVE_CYCLES" in v:
            w = v["SQ_WAVE_CYCLES"]["mean_per_dispatch"]
            print("    wave cycles: wait_any %.2f  wait_inst_any %.2f  active_inst_any %.2f | valu %.2f lds %.2f vmem %.2f wait_inst_lds %.2f" % tuple(
                v[c]["mean_per_dispatch"] / w for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                                                        "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS")))
PY
