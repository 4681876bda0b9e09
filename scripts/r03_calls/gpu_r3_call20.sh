#!/bin/bash
# row-band 128 -> 128 conv kernel: tests, micro-benchmark, end-to-end A/B in one box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "c128_band" 2>&1 | tail -4
timeout 300 python scripts/bench_conv128.py 86 256 2>&1 | grep -v amdgpu.ids
for k in 1 0 1 0; do
FVIT_TUNE_conv_band=$k timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary --no-modes --prof-steps 2 > gpurun_out/r3c20_tmp.json 2>> gpurun_out/r3c20.err
python - $k <<'PY'
import json, sys
d = json.load(open('gpurun_out/r3c20_tmp.json'))
print("conv_band", sys.argv[1], d['ms_per_step'], 'ms/step', d['value'], 'img/s', d['parity']['logits_max_abs_err'])
for r in d['roofline_shapes'][:14]:
    if 'conv3x3' in r['kernel']:
        print(f"   {r['kernel']:30s} wg={r['workgroups']:5d} n={r['launches_per_step']} us={r['avg_launch_us']:7.2f} frac={r['frac']}")
PY
done
grep -v amdgpu.ids gpurun_out/r3c20.err | tail -5
