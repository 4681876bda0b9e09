#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_determinism.py -q -m gpu -k "attn_block_fused or repeatable" 2>&1 | tail -2
for i in 1 2; do
timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary --no-modes --prof-steps 2 > gpurun_out/r3c18_tmp.json 2>> gpurun_out/r3c18.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r3c18_tmp.json'))
print("bench", d['ms_per_step'], 'ms/step', d['value'], 'img/s', d['parity']['logits_max_abs_err'])
for r in d['roofline_shapes'][:12]:
    if 'attnblk' in r['kernel']:
        print(f"   {r['kernel']:30s} wg={r['workgroups']:5d} us={r['avg_launch_us']:7.2f} frac={r['frac']}")
PY
done
