#!/bin/bash
# usage: tools/r3_cfg.sh <cfg> "ENV.." ["ENV.."...] — build / join ms of one configuration (tools/prof_step.py, 3 steps) under several environments
CFG=$1; shift
for envs in "$@"; do
  echo "[$CFG | $envs] $(env $envs timeout -k 10 300 python tools/prof_step.py $CFG 8 2>&1 | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('build min %.3f last %.3f  join min %.3f' % (d['build_ms_min'], d['build_ms'], d['join_ms_min']))")"
done
