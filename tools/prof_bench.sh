#!/bin/bash
# usage: tools/prof_bench.sh <tag> [extra bench args]  — rocprofv3 kernel stats + HBM traffic counters of `python3 bench.py`
set -e
TAG=$1; shift || true
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps 5 --warmup 2 --cpu-sample 0 --other-configs= --profile-steps 0 $@"
echo "[prof] kernel trace"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
echo "[prof] FETCH_SIZE"; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
echo "[prof] WRITE_SIZE"; rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/bench_write.json 2> $OUT/write.err
if [ -z "$KSP_PROF_LIGHT" ]; then
echo "[prof] pmc1"; rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc1 -- python3 $ARGS > /dev/null 2> $OUT/pmc1.err
echo "[prof] pmc2"; rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD --output-format csv -d $OUT/pmc2 -- python3 $ARGS > /dev/null 2> $OUT/pmc2.err
fi
python3 $R/tools/pmc_summary.py $OUT all > $OUT/summary.txt 2>&1 || true
python3 $R/tools/traffic_json.py $OUT > $OUT/traffic.json 2> $OUT/traffic.err || true
# keep only the small artefacts
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
du -sh $OUT
