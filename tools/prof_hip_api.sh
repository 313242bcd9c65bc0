#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_hip_api.sh TAG -> HIP runtime + memory-copy trace of two K4 steps (no counters)
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$1
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --hip-runtime-trace --memory-copy-trace --output-format csv -d $OUT/t -- python3 $GRAFT_REPO_ROOT/tools/k4_step.py 2 > $OUT/log.txt 2>&1
python3 - <<PY
import csv,glob,collections
root="$OUT"
for f in glob.glob(root+"/**/*memory_copy_trace.csv",recursive=True):
    rows=list(csv.DictReader(open(f)))
    print(len(rows), list(rows[0].keys()))
    c=collections.Counter((r.get("Direction"), r.get("Size",r.get("Bytes"))) for r in rows)
    print(c.most_common(15))
for f in glob.glob(root+"/**/*hip_api_trace.csv",recursive=True):
    rows=list(csv.DictReader(open(f)))
    c=collections.Counter(r["Function"] for r in rows)
    print(c.most_common(14))
PY
