#!/bin/bash
# usage: bash scripts/gpu_sweep.sh <tag> "<bench args>" "ENV1=.. ENV2=.." "ENV3=.." ...   (one bench run per env set; "-" = no env)
cd $GRAFT_REPO_ROOT
T=${1:-sw}; shift
ARGS="$1"; shift
mkdir -p gpurun_out
S=gpurun_out/${T}_sweep.log
: > $S
for E in "$@"; do
  if [ "$E" = "-" ]; then EV=""; else EV="$E"; fi
  R=$(env $EV timeout 300 python bench.py --no-cpu-baseline --no-secondary --prof-steps 1 $ARGS 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')
  echo "$E | $ARGS -> $R" >> $S
done
cat $S
