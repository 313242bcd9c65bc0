#!/bin/bash
# usage (GPU box, repo root): bash tools/bucket_sweep.sh "16 13 26" -- per-rank workloads of 1/4/8 shards (class windows) at several buckets
for b in $1; do
  for nc in "50000 200" "25000 100" "12500 50" "6250 25"; do
    set -- $nc
    timeout -k 10 200 python bench.py --bucket $b --bank $1 --classes $2 --steps 832 --warmup 128 --no-cpu-baseline --skip-extras 2>/dev/null > /tmp/bs.json
    python - <<PY
import json
d=json.load(open("/tmp/bs.json")); r=d["roofline"]
print("bucket", $b, "N", d["config"]["N_support"], "value", round(d["value"]), "us/step", round(d["ms_per_step"]*1e3,2), "kernel", round(r["kernel_us"],1), "launch", round(r["launch_us"],1), flush=True)
PY
  done
done
