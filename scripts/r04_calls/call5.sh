#!/bin/bash
# r04 call 5: two-term weights in the Downsample.reduction convs of the deploy plan: kernel test, deploy parity tests, throughput / parity A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_x3.py tests/test_gpu_parity.py tests/test_gpu_backward.py tests/test_gpu_determinism.py -q -m gpu -s -k "conv3x3_two or deploy or whole_model or determin or repeatab" 2>&1 | grep -E "conv two-term|deploy-mode|passed|failed|Error|assert|whole-model" | cut -c1-200 | tail -40
ab() {
  env $1 timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary --prof-steps 1 > gpurun_out/r4c5_ab.json 2>> gpurun_out/r4c5_ab.err
  python - "$1" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/r4c5_ab.json').read().strip().splitlines()[-1])
print(f"{sys.argv[1]:28s} {d['ms_per_step']:.4f} ms/step {d['value']:.0f} img/s  f16 {d['parity']['logits_max_abs_err']}  bf16 {d['parity_bf16']['logits_max_abs_err']}  f16x2 {d['parity_f16x2']['logits_max_abs_err']}  bf16x2 {d['parity_bf16x2']['logits_max_abs_err']}")
PY
}
ab FVIT_DOWN_WEIGHT_TERMS=2
ab FVIT_DOWN_WEIGHT_TERMS=1
ab FVIT_DOWN_WEIGHT_TERMS=2
ab FVIT_DOWN_WEIGHT_TERMS=1
tail -3 gpurun_out/r4c5_ab.err
