set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_dn -- python3 $R/tools/dn_eval_time.py densenet121 64 > $R/gpurun_out/trace_dn.log 2>&1
tail -2 $R/gpurun_out/trace_dn.log
python3 $R/tools/kstats.py $R/gpurun_out/trace_dn | head -24
