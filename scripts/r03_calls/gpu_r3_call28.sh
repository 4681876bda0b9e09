#!/bin/bash
# 8-wave carrier-token kernel (ct_variant = 3): tests, micro-benchmark, end-to-end A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_precision_modes.py -q -m gpu -k "ct_block" -x 2>&1 | tail -4
timeout 300 python scripts/bench_ctblk.py 86 2>&1 | grep -v amdgpu.ids
for k in "0 3" "3 3" "3 2" "0 3" "3 3" "0 3" "3 3"; do
set -- $k
FVIT_TUNE_ct_variant=$1 FVIT_TUNE_ct8_depth=$2 timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary --no-modes --prof-steps 2 > gpurun_out/r3c28_tmp.json 2>> gpurun_out/r3c28.err
python - "$k" <<'PY'
import json, sys
d = json.load(open('gpurun_out/r3c28_tmp.json'))
print("ct_variant/depth", sys.argv[1], d['ms_per_step'], 'ms/step', d['value'], 'img/s', d['parity']['logits_max_abs_err'])
for r in d['roofline_shapes'][:14]:
    if 'ctblk' in r['kernel']:
        print(f"   {r['kernel']:30s} wg={r['workgroups']:5d} n={r['launches_per_step']} us={r['avg_launch_us']:7.2f} frac={r['frac']}")
PY
done
grep -v amdgpu.ids gpurun_out/r3c28.err | tail -5
