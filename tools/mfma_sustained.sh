#!/bin/bash
# tools/mfma_sustained + rocm-smi samples (power, sclk) every ~0.15 s; run on the GPU box: tools/mfma_sustained.sh [seconds]
cd "$(dirname "$0")/.."
out=${2:-gpurun_out/mfma_sustained}
mkdir -p "$out"
( while true; do echo "$(date +%s.%N) $(rocm-smi -P -c --json 2>/dev/null | tr -d '\n')"; sleep 0.1; done ) > "$out/smi.jsonl" &
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
        ts, js = ln.split(" ", 1)
        c = json.loads(js).get("card0", {})
        p = float(str(c.get("Current Socket Graphics Package Power (W)", "nan")))
        f = float(str(c.get("sclk clock speed:", "(nan")).strip("()Mhz "))
        smi.append((float(ts), p, f))
    except Exception:
        pass
print("\n".join(txt for _, txt in marks))
# per mode: the samples between its START and END marks (the first half second left out: the power loop settles)
starts = {txt.split(" START")[0]: t for t, txt in marks if txt.endswith("START")}
for t1, txt in marks:
    if " END" not in txt:
        continue
    name = txt.split(" END")[0]
    t0 = starts.get(name)
    sel = [(p, f) for ts, p, f in smi if t0 is not None and t0 + 0.5 <= ts <= t1]
    if sel:
        print("%-40s rocm-smi: %3d samples, power mean %.0f W max %.0f W, sclk mean %.0f MHz" % (
            name, len(sel), sum(p for p, _ in sel) / len(sel), max(p for p, _ in sel), sum(f for _, f in sel) / len(sel)))
json.dump({"marks": marks, "smi": smi}, open(out + "/summary.json", "w"))
PY
