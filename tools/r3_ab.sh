#!/bin/bash
# usage: tools/r3_ab.sh <steps> "ENV.." ["ENV.."...] — ms per C2 step (bench.py, no CPU baseline, no other configs) under several environments, in turn
STEPS=$1; shift
for rep in 1 2; do
for envs in "$@"; do
  env $envs timeout -k 10 300 python bench.py --steps $STEPS --warmup 10 --cpu-sample 0 --other-configs= --profile-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$envs] ms_per_step %.4f build %.4f join %.4f other %.4f' % (d['ms_per_step'], d['stage_ms']['build_blocks'], d['stage_ms']['join'], d['ms_per_step']-d['stage_ms']['build_blocks']-d['stage_ms']['join']))"
done
done
