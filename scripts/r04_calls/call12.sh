#!/bin/bash
# r04 call 12: does a settle phase after the graph capture change the driver-form measurement (--steps 20 --warmup 5)?
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for s in 0 30; do
  FVIT_BENCH_SETTLE=$s python bench.py --steps 20 --warmup 5 --no-secondary --no-modes --no-cpu-baseline --prof-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('settle $s steps 20:', d['ms_per_step'], d['value'])"
done
done
FVIT_BENCH_SETTLE=0 python bench.py --steps 50 --warmup 10 --no-secondary --no-modes --no-cpu-baseline --prof-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('settle 0 steps 50:', d['ms_per_step'], d['value'])"
FVIT_BENCH_SETTLE=0 python bench.py --steps 200 --warmup 10 --no-secondary --no-modes --no-cpu-baseline --prof-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('settle 0 steps 200:', d['ms_per_step'], d['value'])"
