#!/bin/bash
# r04 call 9: two-term weights inside attnblk (the stage-2 window attention of the x2 operand modes): kernel test, x2-mode model tests, images/s per mode A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_precision_modes.py -q -m gpu -s -k "attn_block_two or meet_the_bar or repeatable" 2>&1 | grep -E "passed|failed|Error|assert|logits max-abs" | cut -c1-220 | tail -20
ab() {
  E=$1; shift
  env $E timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary --no-modes --prof-steps 1 "$@" > gpurun_out/r4c9_ab.json 2>> gpurun_out/r4c9_ab.err
  python - "$E $*" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/r4c9_ab.json').read().strip().splitlines()[-1])
r = d.get('roofline') or {}
print(f"{sys.argv[1][:60]:60s} {d['ms_per_step']:.4f} ms/step {d['value']:.0f} img/s err {d['parity']['logits_max_abs_err']} dom {r.get('kernel')} {r.get('frac')} {r.get('avg_launch_us')}")
PY
}
for op in bf16x2 f16x2; do
ab FVIT_TUNE_attn_fused_x2=1 --operand $op
ab FVIT_TUNE_attn_fused_x2=0 --operand $op
done
ab FVIT_TUNE_attn_fused_x2=1 --operand bf16x2 --streams 3 --join-from 0
ab X=0
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_detail.json'))
PY
tail -3 gpurun_out/r4c9_ab.err
