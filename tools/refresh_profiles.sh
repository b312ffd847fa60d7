#!/bin/bash
# Refresh the committed evidence of a round on the GPU box (run through gpurun from the repo root):
#     tools/refresh_profiles.sh r02
# writes gpurun_out/<tag>/...; copy what is to be judged into profiles/ afterwards (tools/collect_profiles.py).
set -e
tag=${1:-r03}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
echo "== kernel stats: rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu" | tee "$out/commands.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o s -- python3 bench.py --steps 5 --warmup 2 --no-cpu > "$out/kernel_stats_bench.json" 2> "$out/kernel_stats.log"
find "$out/stats" -name "*kernel_trace.csv" -delete
cp "$(find "$out/stats" -name "*kernel_stats.csv" | head -1)" "$out/kernel_stats.csv"
echo "== PMC passes" | tee -a "$out/commands.txt"
tools/pmc_passes.sh "$out/pmc" --steps 1 --warmup 1 --no-cpu
cat "$out/pmc/commands.txt" >> "$out/commands.txt"
echo "== bench lines" | tee -a "$out/commands.txt"
python3 bench.py > "$out/bench_C.json" 2> "$out/bench_C.log"
python3 bench.py --n 1616 --eps 0.044 --no-cpu > "$out/ragged_bench.json" 2> "$out/ragged.log"
python3 bench.py --config E --no-cpu --steps 3 --warmup 1 > "$out/config_E_fp32_bench.json" 2> "$out/E.log"
python3 bench.py --config D --patches 1024 --nq 4194304 --no-cpu --steps 3 --warmup 1 > "$out/config_D_one_gpu_bench.json" 2> "$out/D.log"
# the HBM-bound kernel-matrix build of config E on its own (the default fit evaluates the tiles inside the factorisation)
PMK_FUSE_K1=0 python3 bench.py --config E --no-cpu --no-e2e --steps 3 --warmup 1 > "$out/config_E_fp32_unfused_bench.json" 2> "$out/E_unfused.log"
SP_JSON="$out/single_problem.json" python3 tools/single_problem.py > "$out/single_problem.log" 2>&1
python3 tools/e2e_breakdown.py > "$out/e2e_breakdown.txt" 2>&1
tail -c 600 "$out/bench_C.json"
