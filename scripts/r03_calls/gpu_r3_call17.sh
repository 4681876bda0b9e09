#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_precision_modes.py tests/test_gpu_kernels.py -q -m gpu -k "mlp_fused or win_mlp or wide_row" 2>&1 | tail -3
timeout 300 python scripts/timeline_winmlp.py 2>&1 | grep -v amdgpu | grep -A16 "M=4214\|M=18240" | grep -v COLD -A0 | head -40
for i in 1 2; do
timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary --no-modes --prof-steps 2 > gpurun_out/r3c17_tmp.json 2>> gpurun_out/r3c17.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r3c17_tmp.json'))
print("bench", d['ms_per_step'], 'ms/step', d['value'], 'img/s')
for r in d['roofline_shapes'][:8]:
    if 'winmlp' in r['kernel']:
        print(f"   {r['kernel']:30s} wg={r['workgroups']:5d} us={r['avg_launch_us']:7.2f} frac={r['frac']}")
PY
done
