#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_backbone_pmc.sh TAG k2|k4|k5 [steps]
# rocprofv3 kernel trace + stats, then PMC passes (never combined with a trace) of the K2 predict calls (folded channels_last
# ResNet-18 + head) or of DenseNet-121 training steps (K4); condense with tools/pmc_by_kernel.py
set -e
TAG=$1; WHAT=$2; STEPS=${3:-4}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
if [ "$WHAT" = k4 ]; then
  SCRIPT=$GRAFT_REPO_ROOT/tools/k4_step.py
elif [ "$WHAT" = k5 ]; then
  SCRIPT=$GRAFT_REPO_ROOT/tools/k5_time.py
else
  SCRIPT=$GRAFT_REPO_ROOT/tools/k2_step.py
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $SCRIPT $STEPS > $OUT/trace.log 2>&1
echo trace done; tail -1 $OUT/trace.log
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1 -- python3 $SCRIPT $STEPS > $OUT/pmc1.log 2>&1
echo pmc1 done
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -- python3 $SCRIPT $STEPS > $OUT/pmc2.log 2>&1
echo pmc2 done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 $SCRIPT $STEPS > $OUT/pmc3.log 2>&1
echo pmc3 done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -- python3 $SCRIPT $STEPS > $OUT/pmc4.log 2>&1
echo pmc4 done
python3 $GRAFT_REPO_ROOT/tools/pmc_by_kernel.py $OUT $OUT/by_kernel.json > $OUT/by_kernel.txt 2>&1 || true
head -30 $OUT/by_kernel.txt
