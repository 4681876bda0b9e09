#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_precision_modes.py tests/test_gpu_kernels.py -q -m gpu -k "wide_row or mlp_fused or two_weight_terms" 2>&1 | tail -4
for f in 3 2 3 2; do
FVIT_TUNE_win_mlp256=$f timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary --no-modes --prof-steps 2 > gpurun_out/r3c14_tmp.json 2>> gpurun_out/r3c14.err
python - $f <<'PY'
import json, sys
d = json.load(open('gpurun_out/r3c14_tmp.json'))
print("win_mlp256 =", sys.argv[1], "bench", d['ms_per_step'], 'ms/step', d['value'], 'img/s', 'parity', d['parity']['logits_max_abs_err'])
for r in d['roofline_shapes'][:24]:
    if 'winmlp_kernel<256' in r['kernel']:
        print(f"   {r['kernel']:30s} wg={r['workgroups']:5d} n={r['launches_per_step']:3d} us={r['avg_launch_us']:7.2f} frac={r['frac']}")
PY
done
tail -2 gpurun_out/r3c14.err
