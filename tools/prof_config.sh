#!/bin/bash
# usage: tools/prof_config.sh <tag> <config> [steps]  — rocprofv3 kernel stats + HBM traffic counters of tools/prof_step.py on one
# configuration; the summaries (top kernels, per-kernel FETCH / WRITE) land in gpurun_out/pc_<tag>/
TAG=$1; CFG=$2; STEPS=${3:-3}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
DATA=/tmp/prof_$CFG.npz
python3 $R/tools/prof_step.py $CFG 0 dump $DATA
echo "[prof] $CFG kernel trace"
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/prof_step.py $CFG $STEPS load $DATA > $OUT/run_trace.log 2> $OUT/trace.err
tail -1 $OUT/run_trace.log
echo "[prof] $CFG FETCH_SIZE"
timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/prof_step.py $CFG 2 load $DATA 2> $OUT/fetch.err | grep --line-buffered -v "^{" 
echo "[prof] $CFG WRITE_SIZE"
timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/tools/prof_step.py $CFG 2 load $DATA 2> $OUT/write.err | grep --line-buffered -v "^{"
rm -f $DATA
python3 $R/tools/config_summary.py $OUT $CFG > $OUT/summary.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*counter_collection.csv" -size +20M -delete
tail -40 $OUT/summary.txt
