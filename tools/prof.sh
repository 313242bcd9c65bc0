#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/prof.sh TAG B N d C
# kernel-trace + stats in one run, PMC counters in separate runs (never combined with traces).
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py "$@" 30 > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/pmc1 -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py "$@" 5 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py "$@" 5 > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py "$@" 5 > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py "$@" 5 > $OUT/pmc4.log 2>&1
find $OUT -name "*.csv" | head -30
