#!/bin/bash
# r04 call 10: two-term attnblk with the slices of the next head requested one phase earlier (no memory wait at barrier A2); the training smoke test
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_precision_modes.py tests/test_gpu_backward.py -q -m gpu -s -k "attn_block_two or meet_the_bar or repeatable or training_steps" 2>&1 | grep -E "passed|failed|Error|assert|logits max-abs|training losses" | cut -c1-220 | tail -12
ab() {
  E=$1; shift
  env $E timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary --no-modes --prof-steps 1 "$@" > gpurun_out/r4c10_ab.json 2>> gpurun_out/r4c10_ab.err
  python - "$E $*" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/r4c10_ab.json').read().strip().splitlines()[-1])
det = json.load(open('gpurun_out/bench_detail.json'))
ab = [r for r in det['roofline_shapes'] if r['kernel'].startswith('attnblk')]
print(f"{sys.argv[1][:50]:50s} {d['ms_per_step']:.4f} ms/step {d['value']:.0f} img/s err {d['parity']['logits_max_abs_err']}  attnblk {[(r['kernel'], r['workgroups'], r['avg_launch_us']) for r in ab[:2]]}")
PY
}
ab X=0 --operand bf16x2
ab X=0 --operand f16x2
ab FVIT_TUNE_attn_fused_x2=0 --operand bf16x2
ab X=0
tail -2 gpurun_out/r4c10_ab.err
