#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_train.sh TAG [NW_NHWC_TRAINING]  -> kernel trace of DenseNet-121 training steps (42 @224)
set -e
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp NW_NHWC_TRAINING=${2:-1}
cd /tmp
# an un-traced run first: the first weight gradient of the strided stem makes MIOpen search its solvers (5 s of naive /
# Tensile / CK kernels in the first step of a process with an empty user find-db); the trace below is of a process that finds
# the result in ~/.config/miopen
python3 $GRAFT_REPO_ROOT/tools/k4_step.py 1 > $OUT/warm.log 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/k4_step.py > $OUT/trace.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/kstats.py $OUT/trace > $OUT/stats_top.txt 2>&1 || true
head -45 $OUT/stats_top.txt
tail -3 $OUT/trace.log
