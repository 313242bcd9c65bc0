#!/bin/bash
# usage (GPU box, repo root): bash tools/ab_lib.sh tools/lib_old.so [rounds] -- bench.py's K3 launch with the given library and
# with the in-tree one, alternated on the same device (kernel_us = tile kernel, HIP events inside the library)
old=$1; n=${2:-3}
for i in $(seq $n); do
  for which in old new; do
    if [ $which = old ]; then export NW_HIP_LIB=$old; else unset NW_HIP_LIB; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline --skip-extras 2>/dev/null > /tmp/ab.json || exit 1
    python - <<PY
import json
d=json.load(open("/tmp/ab.json")); r=d["roofline"]
print("$which", "value", round(d["value"]), "kernel_us", round(r["kernel_us"],1), "launch_us", round(r["launch_us"],1), "frac", round(r["frac"],4), flush=True)
PY
  done
done
