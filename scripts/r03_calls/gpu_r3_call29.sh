#!/bin/bash
# 8-wave carrier-token kernel as the default: kernel / determinism / parity tests, micro-benchmark + timeline, end-to-end A/B against the 4-wave form
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_precision_modes.py tests/test_gpu_determinism.py tests/test_gpu_parity.py -q -m gpu -k "ct_block or knobs or repeatable or fvit0 or bench_configuration or poison" -x 2>&1 | tail -4
timeout 300 python scripts/bench_ctblk.py 86 2>&1 | grep -v amdgpu.ids | tail -22
for k in 0 3 0 3; do
FVIT_TUNE_ct_variant=$k timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary --no-modes --prof-steps 2 > gpurun_out/r3c29_tmp.json 2>> gpurun_out/r3c29.err
python - "$k" <<'PY'
import json, sys
d = json.load(open('gpurun_out/r3c29_tmp.json'))
print("ct_variant", sys.argv[1], d['ms_per_step'], 'ms/step', d['value'], 'img/s', d['parity']['logits_max_abs_err'], [ (r['avg_launch_us']) for r in d['roofline_shapes'] if 'ctblk' in r['kernel']])
PY
done
grep -v amdgpu.ids gpurun_out/r3c29.err | tail -5
