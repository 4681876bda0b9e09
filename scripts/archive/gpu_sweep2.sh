#!/bin/bash
# usage: bash scripts/gpu_sweep2.sh <tag> -- runs a fixed list of "bench-args | env" combinations (one per line in $2 file)
cd $GRAFT_REPO_ROOT
T=${1:-sw}
mkdir -p gpurun_out
S=gpurun_out/${T}_sweep.log
: > $S
while IFS='|' read -r ARGS EV; do
  [ -z "$ARGS$EV" ] && continue
  R=$(env $EV timeout 300 python bench.py --no-cpu-baseline --no-secondary --prof-steps 1 $ARGS 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')
  echo "$ARGS | $EV -> $R" >> $S
done < $2
cat $S
