#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/prof_r04.sh [bench] [T] [k2] [k4] [k5]   (default: all)
# The round's rocprofv3 evidence: kernel traces with stats, then PMC passes (never combined with traces).
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
WHAT="${@:-bench T k2 k4 k5}"
for w in $WHAT; do
  case $w in
    bench) bash $R/tools/prof_bench.sh bench_r04 > $OUT/prof_bench_r04.log 2>&1 || true ;;
    T)     bash $R/tools/prof.sh T_r04 256 10000 512 200 fwd > $OUT/prof_T_r04.log 2>&1 || true ;;
    k2)    bash $R/tools/prof_backbone_pmc.sh r04_k2 k2 10 > $OUT/prof_k2_r04.log 2>&1 || true ;;
    k4)    bash $R/tools/prof_backbone_pmc.sh r04_k4 k4 3 > $OUT/prof_k4_r04.log 2>&1 || true ;;
    k5)    bash $R/tools/prof_backbone_pmc.sh r04_k5 k5 > $OUT/prof_k5_r04.log 2>&1 || true ;;
  esac
  echo $w done
done
