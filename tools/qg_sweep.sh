#!/bin/bash
# usage (GPU box, repo root): bash tools/qg_sweep.sh "8 13 26 52" -- query tiles per group of the persistent kernel (NW_QG)
for g in $1; do
  NW_QG=$g timeout -k 10 200 python bench.py --steps 520 --warmup 128 --no-cpu-baseline --skip-extras 2>/dev/null > /tmp/qg.json
  python - <<PY
import json
d=json.load(open("/tmp/qg.json")); r=d["roofline"]
print("NW_QG", $g, "value", round(d["value"]), "kernel", round(r["kernel_us"],1), "launch", round(r["launch_us"],1), flush=True)
PY
done
