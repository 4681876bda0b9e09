#!/bin/bash
# r04 call 1: the compact bench line on the box (what the driver will parse) + A/B of joining the stream shards in front of stage 3
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 600 python bench.py ) > gpurun_out/r4c1_bench_default.json 2> gpurun_out/r4c1_bench_default.err
tail -1 gpurun_out/r4c1_bench_default.json | wc -c
tail -1 gpurun_out/r4c1_bench_default.json
tail -4 gpurun_out/r4c1_bench_default.err
cp gpurun_out/bench_detail.json gpurun_out/r4c1_bench_detail.json
ab() {
  timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary --no-modes --prof-steps 1 "$@" > gpurun_out/r4c1_ab.json 2>> gpurun_out/r4c1_ab.err
  python - "$*" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/r4c1_ab.json').read().strip().splitlines()[-1])
print(f"{sys.argv[1]:40s} {d['ms_per_step']:.4f} ms/step {d['value']:.0f} img/s err {d['parity']['logits_max_abs_err']}")
PY
}
ab
ab --join-from 3
ab --join-from 3 --streams 2
ab --join-from 3 --streams 4
ab --join-from 2
ab --streams 4
ab
ab --join-from 3
tail -5 gpurun_out/r4c1_ab.err
