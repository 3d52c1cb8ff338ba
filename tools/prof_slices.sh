#!/bin/bash
# usage: tools/prof_slices.sh <tag> <N> <G>  -- kernel trace of tools/probe_slices.py (sharded stage 1 on one GPU)
set -e
TAG=$1; N=${2:-28284}; G=${3:-8}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/probe_slices.py $N $G > $OUT/trace.log 2>&1
