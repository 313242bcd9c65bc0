#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/prof_r02.sh
# All rocprofv3 evidence of the round: kernel traces with stats, then PMC passes (never combined with traces).
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
# 1. the bench workload (K3): trace + counters
bash $R/tools/prof_bench.sh bench_r02 > $OUT/prof_bench_r02.log 2>&1 || true
# 2. the T forward: trace + counters
bash $R/tools/prof.sh T_r02 256 10000 512 200 fwd > $OUT/prof_T_r02.log 2>&1 || true
# 3. the other kernels: trace only, plus HBM counters for the influence kernel
mkdir -p $OUT/prof_kern_r02
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_kern_r02/trace -- python3 $R/tools/prof_kernels.py 20 > $OUT/prof_kern_r02/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_kern_r02/pmc3 -- python3 $R/tools/prof_kernels.py 5 > $OUT/prof_kern_r02/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_kern_r02/pmc4 -- python3 $R/tools/prof_kernels.py 5 > $OUT/prof_kern_r02/pmc4.log 2>&1
# 4. DenseNet-121 folded forward (the fused 1x1 convolutions)
mkdir -p $OUT/prof_dn_r02
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_dn_r02 -- python3 $R/tools/dn_prof.py 3 > $OUT/prof_dn_r02.log 2>&1
echo profiles done
