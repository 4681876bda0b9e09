#!/bin/bash
# stage 3 as one launch of persistent per-window workgroups (win_stage3): parity / repeatability with the knob on, end-to-end A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
FVIT_TUNE_win_stage3=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_determinism.py -q -m gpu -k "fvit0 or bench_configuration or repeatable or stage_maps or poison" -x 2>&1 | tail -4
for k in 0 1 0 1; do
FVIT_TUNE_win_stage3=$k timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary --no-modes --prof-steps 2 > gpurun_out/r3c30_tmp.json 2>> gpurun_out/r3c30.err
python - "$k" <<'PY'
import json, sys
d = json.load(open('gpurun_out/r3c30_tmp.json'))
print("win_stage3", sys.argv[1], d['ms_per_step'], 'ms/step', d['value'], 'img/s', d['parity']['logits_max_abs_err'])
for r in d['roofline_shapes'][:12]:
    if 'stage3' in r['kernel'] or 'winblk' in r['kernel'] or '<512>' in r['kernel']:
        print(f"   {r['kernel']:30s} wg={r['workgroups']:5d} n={r['launches_per_step']} us={r['avg_launch_us']:7.2f} frac={r['frac']}")
PY
done
grep -v amdgpu.ids gpurun_out/r3c30.err | tail -5
