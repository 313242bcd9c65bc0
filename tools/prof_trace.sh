#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_trace.sh TAG script.py [args] -> kernel trace + per-kernel totals of a python script
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
S=$GRAFT_REPO_ROOT/$1; shift
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $S "$@" > $OUT/trace.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_by_kernel.py $OUT $OUT/by_kernel.json > $OUT/by_kernel.txt 2>&1 || true
grep -v Warn $OUT/trace.log | grep "ms per" || true
head -30 $OUT/by_kernel.txt
