#!/bin/bash
# usage: tools/exp_env.sh VAR "v1 v2 ..." [config] [reps]  — same-box A/B of one environment knob: build / join ms of tools/time_build.py
VAR=$1; VALS=$2; CFG=${3:-C2}; REPS=${4:-6}
for v in $VALS $VALS; do
  env $VAR=$v python3 tools/time_build.py $CFG $REPS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', 'build', round(d['build_ms'],3), 'join', round(d['join_ms'],3), 'wgs', d['wgs'], 'active', d['active'])"
done
