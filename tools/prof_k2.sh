#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_k2.sh TAG -> kernel trace + stats of the K2 predict calls (folded channels_last ResNet-18 + head)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$1
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/k2_step.py 30 > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log
