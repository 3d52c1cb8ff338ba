#!/bin/bash
# usage: tools/prof_timeline.sh <tag>  — kernel trace of a short bench run, then the dispatch timeline of its last step (tools/gaps.py)
set -e
TAG=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/tl_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 6 --warmup 2 --cpu-sample 0 --other-configs= --profile-steps 0 > $OUT/bench.json 2> $OUT/trace.err
python3 $R/tools/gaps.py $OUT/trace > $OUT/timeline.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*memory_copy*.csv" -delete
tail -3 $OUT/timeline.txt
