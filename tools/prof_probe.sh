#!/bin/bash
# usage: tools/prof_probe.sh <tag> <N>
set -e
TAG=$1; N=${2:-10000}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/probe.py $N > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc1 -- python3 $R/tools/probe.py $N > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD --output-format csv -d $OUT/pmc2 -- python3 $R/tools/probe.py $N > $OUT/pmc2.log 2>&1
rocprofv3 --pmc TA_BUSY TA_TOTAL_WAVEFRONTS TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ --output-format csv -d $OUT/pmc3 -- python3 $R/tools/probe.py $N > $OUT/pmc3.log 2>&1 || true
find $OUT -name "*.csv" | head -20
