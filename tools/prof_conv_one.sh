#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_conv_one.sh TAG n cin h w cout k -> kernel stats + a few SQ counters of one convolution shape
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/conv_one.py "$@" > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/pmc1 -- python3 $GRAFT_REPO_ROOT/tools/conv_one.py "$@" > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc2 -- python3 $GRAFT_REPO_ROOT/tools/conv_one.py "$@" > $OUT/pmc2.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "nw_conv_nhwc_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:28s} {sum(v) / len(v):16.1f}")
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "nw_conv_nhwc_kernel" in r["Name"]:
            print(r["Name"][:60], "calls", r["Calls"], "avg us", float(r["AverageNs"]) / 1e3)
PY
