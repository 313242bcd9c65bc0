#!/bin/bash
# usage (GPU box, repo root): bash tools/bd_pmc.sh rows c ctot -- HBM bytes of the two passes of nw_bn_dgrad1x1_bwd_f16x2 (tools/bench_bn_dgrad)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/bd_pmc
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p1 -- $GRAFT_REPO_ROOT/tools/bench_bn_dgrad $1 $2 128 $3 > $OUT/p1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p2 -- $GRAFT_REPO_ROOT/tools/bench_bn_dgrad $1 $2 128 $3 > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/p3 -- $GRAFT_REPO_ROOT/tools/bench_bn_dgrad $1 $2 128 $3 > $OUT/p3.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("p1", "p2", "p3"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(d, k, {c: round(sum(x) / len(x), 1) for c, x in v.items()})
PY
