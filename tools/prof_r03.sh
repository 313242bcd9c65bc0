#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/prof_r03.sh
# The round's rocprofv3 evidence: kernel traces with stats, then PMC passes (never combined with traces).
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
bash $R/tools/prof_bench.sh bench_r03 > $OUT/prof_bench_r03.log 2>&1 || true
echo bench done
bash $R/tools/prof.sh T_r03 256 10000 512 200 fwd > $OUT/prof_T_r03.log 2>&1 || true
echo T done
bash $R/tools/prof_k2.sh K2_r03 > $OUT/prof_K2_r03.log 2>&1 || true
echo K2 done
bash $R/tools/prof_train.sh K4_r03 1 > $OUT/prof_K4_r03.log 2>&1 || true
echo K4 done
