#!/bin/bash
# usage: tools/r3_quick.sh <tag> [pytest -k expression]  — smoke, a slice of the engine tests, a short C2 bench with the per-phase table
TAG=$1; K=${2:-"default or pass_by_pass"}
O=gpurun_out/q_$TAG; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout -k 10 400 python -m pytest tests/test_engine_gpu.py tests/test_partition_gpu.py -x -q -k "$K" > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --other-configs= > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.err
python - <<EOF
import json
d=json.load(open("$O/bench.json"))
print("ms_per_step", round(d["ms_per_step"],4), {k: round(v,4) for k,v in d["stage_ms"].items() if isinstance(v,float)}, "edges", d["config"]["nonzero_pairs"], "chk", d["config"]["checksum"])
for g in d["roofline"]["groups"]: print("  ", g["group"], round(g["ms"],4))
EOF
