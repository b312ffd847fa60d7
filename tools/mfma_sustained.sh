#!/bin/bash
# tools/mfma_sustained + rocm-smi samples (power, sclk) every ~0.15 s; run on the GPU box: tools/mfma_sustained.sh [seconds]
cd "$(dirname "$0")/.."
out=${2:-gpurun_out/mfma_sustained}
mkdir -p "$out"
( while true; do rocm-smi -P -c --json 2>/dev/null | tr -d '\n'; echo; sleep 0.1; done ) > "$out/smi.jsonl" &
smi=$!
tools/mfma_sustained "${1:-4}" | while read -r line; do echo "$(date +%s.%N) $line"; done > "$out/run.log"
kill $smi
python3 - "$out" <<'PY'
import json, sys
out = sys.argv[1]
run = [l.split(" ", 1) for l in open(out + "/run.log").read().strip().split("\n")]
marks = [(float(t), txt) for t, txt in run]
smi = []
for ln in open(out + "/smi.jsonl"):
    try:
        c = json.loads(ln).get("card0", {})
        p = float(str(c.get("Current Socket Graphics Package Power (W)", "nan")))
        f = float(str(c.get("sclk clock speed:", "(nan")).strip("()Mhz "))
        smi.append((p, f))
    except Exception:
        pass
print("\n".join(txt for _, txt in marks))
print("rocm-smi samples over the whole run: %d, power mean %.0f W max %.0f W, sclk mean %.0f MHz" % (
    len(smi), sum(p for p, _ in smi) / max(1, len(smi)), max([p for p, _ in smi] + [0]), sum(f for _, f in smi) / max(1, len(smi))))
json.dump({"marks": marks, "smi": smi}, open(out + "/summary.json", "w"))
PY
