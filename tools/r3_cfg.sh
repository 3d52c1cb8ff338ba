#!/bin/bash
# usage: tools/r3_cfg.sh <cfg> "ENV.." ["ENV.."...] — build / join ms of one configuration (tools/prof_step.py, 3 steps) under several environments
CFG=$1; shift
for envs in "$@"; do
  echo "[$CFG | $envs] $(env $envs timeout -k 10 300 python tools/prof_step.py $CFG 3 2>&1 | grep '^{' | tail -1)"
done
