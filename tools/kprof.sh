#!/bin/bash
# usage (GPU box, repo root): bash tools/kprof.sh NAME script.py [args]  -> gpurun_out/kprof_NAME/ + medians on stdout
R=$GRAFT_REPO_ROOT; N=$1; shift
export TMPDIR=/tmp; cd /tmp
rm -rf $R/gpurun_out/kprof_$N
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kprof_$N -- python3 $R/"$@" > $R/gpurun_out/kprof_$N.log 2>&1
cd $R; python3 tools/kstats.py gpurun_out/kprof_$N nw_ | head -40
find gpurun_out/kprof_$N -name "*.csv" ! -name "*kernel_trace.csv" -delete
