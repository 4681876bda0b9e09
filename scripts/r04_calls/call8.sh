#!/bin/bash
# r04 call 8: shard-size sweep of the 2-shard + join structure (the box-level noise of call 7 was +-10 images/s)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
ab() {
  timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary --no-modes --prof-steps 0 "$@" > gpurun_out/r4c8_ab.json 2>> gpurun_out/r4c8_ab.err
  python - "$*" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/r4c8_ab.json').read().strip().splitlines()[-1])
print(f"{sys.argv[1][:60]:60s} {d['ms_per_step']:.4f} ms/step {d['value']:.0f} img/s")
PY
}
ab
for s in 132,124 136,120 140,116 144,112 152,104 160,96 124,132 112,144; do ab --shard-sizes $s; done
ab
ab --streams 3 --shard-sizes 96,96,64
ab --streams 3 --shard-sizes 104,88,64
ab --streams 3 --shard-sizes 64,96,96
tail -2 gpurun_out/r4c8_ab.err
