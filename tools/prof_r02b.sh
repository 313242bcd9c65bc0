#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_r02b.sh -- the kernels that changed late in round 2 (backward on split rows,
# convolutions with loader waves): kernel traces with stats, HBM counters for the backward's kernels in separate passes.
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf $OUT/prof_kern_r02b $OUT/prof_dn_r02b
mkdir -p $OUT/prof_kern_r02b
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_kern_r02b/trace -- python3 $R/tools/prof_kernels.py 20 > $OUT/prof_kern_r02b/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_kern_r02b/pmc3 -- python3 $R/tools/prof_kernels.py 5 > $OUT/prof_kern_r02b/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_kern_r02b/pmc4 -- python3 $R/tools/prof_kernels.py 5 > $OUT/prof_kern_r02b/pmc4.log 2>&1
mkdir -p $OUT/prof_dn_r02b
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_dn_r02b -- python3 $R/tools/dn_prof.py 3 > $OUT/prof_dn_r02b.log 2>&1
echo profiles done
