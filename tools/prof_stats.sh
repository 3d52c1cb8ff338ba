#!/bin/bash
# usage: tools/prof_stats.sh <tag> <python script> [args]  — rocprofv3 kernel stats of one python tool; prints the top kernels
set -e
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/ks_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/"$@" > $OUT/run.log 2> $OUT/trace.err
S=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python3 - "$S" <<'PY' > $OUT/top.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:28]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):5d} avg_us {float(r["AverageNs"])/1e3:9.1f} total_ms {float(r["TotalDurationNs"])/1e6:9.2f}')
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
cat $OUT/top.txt
