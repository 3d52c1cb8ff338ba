#!/bin/bash
# usage: tools/prof_trace.sh <tag> <N>   -- kernel trace of tools/probe.py (per-kernel average durations)
set -e
TAG=$1; N=${2:-10000}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/probe.py $N > $OUT/trace.log 2>&1
