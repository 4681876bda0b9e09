#!/bin/bash
# r04 call 3: hat_backward after the ADVICE r03 fixes (scaling, sink, DDP, validation), x3 at block level, repeated A/B of the stage-3 join
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_backward.py -q -m gpu -s 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r4c3_bwd.log; tail -30 gpurun_out/r4c3_bwd.log
timeout 600 python -m pytest tests/test_gpu_x3.py tests/test_gpu_runtime.py -q -m gpu -s -k "blocks_x3 or device_and_mode or long_windows" 2>&1 | grep -E "rel err|passed|failed|Error|assert" | tail -15
ab() {
  timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary --no-modes --prof-steps 1 "$@" > gpurun_out/r4c3_ab.json 2>> gpurun_out/r4c3_ab.err
  python - "$*" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/r4c3_ab.json').read().strip().splitlines()[-1])
r = d.get('roofline') or {}
print(f"{sys.argv[1]:34s} {d['ms_per_step']:.4f} ms/step {d['value']:.0f} img/s err {d['parity']['logits_max_abs_err']}  dominant {r.get('kernel')} frac {r.get('frac')}")
PY
}
for i in 1 2 3; do
ab
ab --join-from 3 --streams 2
ab --join-from 3
done
tail -3 gpurun_out/r4c3_ab.err
