#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/prof_trace.sh TAG B N d C what
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf $OUT/trace; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py "$@" 30 > $OUT/trace.log 2>&1
cat $OUT/trace/*/*_kernel_stats.csv | cut -d, -f1-4,6,7 | cut -c1-220
