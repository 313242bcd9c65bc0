#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_fold.sh TAG ARCH   -> kernel trace + stats of the folded channels_last forward
set -e
TAG=$1; ARCH=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp CL=1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/fold_time.py $ARCH > $OUT/trace.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/kstats.py $OUT/trace > $OUT/stats_top.txt 2>&1 || true
cat $OUT/stats_top.txt
