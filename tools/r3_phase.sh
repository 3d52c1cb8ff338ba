#!/bin/bash
# usage: tools/r3_phase.sh <tag> "ENV1=.. ENV2=.." ["ENV.."...]  — C2 bench phase table under several environments
TAG=$1; shift
O=gpurun_out/p_$TAG; mkdir -p $O
i=0
for envs in "$@"; do
  i=$((i+1))
  env $envs timeout -k 10 200 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --other-configs= > $O/bench_$i.json 2> $O/bench_$i.err || { tail -c 300 $O/bench_$i.err; }
  python - <<EOF
import json
d=json.load(open("$O/bench_$i.json"))
print("[$envs] ms_per_step", round(d["ms_per_step"],4), "build", round(d["stage_ms"]["build_blocks"],4), "join", round(d["stage_ms"]["join"],4), "edges", d["config"]["nonzero_pairs"], " | ".join(f'{g["group"][:14]} {g["ms"]:.3f}' for g in d["roofline"]["groups"]))
EOF
done
