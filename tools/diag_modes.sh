#!/bin/bash
# usage: tools/diag_modes.sh <cfg> <n> <iters>  -- tools/diag_c4.py under each engine mode (mismatch counts)
R=$GRAFT_REPO_ROOT
for m in "" "KSP_KEY_GROUPS=0" "KSP_HASH_GROUP=0" "KSP_NO_SCHED=1" "KSP_DEBUG_COOP=100000" "KSP_COLLECT=0" "KSP_REORDER=0"; do
  echo "== mode: $m"
  env $m timeout -k 10 200 python $R/tools/diag_c4.py $1 $2 $3 2>&1 | grep -v amdgpu.ids | grep "iterations\|MISMATCH" | tail -n 3
done
