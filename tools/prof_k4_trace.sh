#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_k4_trace.sh TAG  -> kernel trace of DenseNet-121 training steps + per-kernel totals
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$1
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python3 $GRAFT_REPO_ROOT/tools/k4_step.py 1 > $OUT/warm.log 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/k4_step.py 3 > $OUT/trace.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_by_kernel.py $OUT $OUT/by_kernel.json > $OUT/by_kernel.txt 2>&1 || true
head -34 $OUT/by_kernel.txt
