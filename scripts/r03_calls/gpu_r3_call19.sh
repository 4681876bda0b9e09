#!/bin/bash
# two-term weights inside the fused stage-3 attention (winblk) and carrier-token (ctblk) kernels: kernel tests, model parity, images/s per mode
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_precision_modes.py -q -m gpu -k "two_weight_terms or meet_the_bar or repeatable or fvit4" 2>&1 | tail -5
for op in bf16x2 f16x2; do
timeout 300 python bench.py --operand $op --steps 40 --warmup 10 --no-cpu-baseline --no-secondary --no-modes --prof-steps 2 > gpurun_out/r3c19_$op.json 2>> gpurun_out/r3c19.err
python - $op <<'PY'
import json, sys
d = json.load(open(f'gpurun_out/r3c19_{sys.argv[1]}.json'))
print(sys.argv[1], d['ms_per_step'], 'ms/step', d['value'], 'img/s', d['parity']['logits_max_abs_err'])
tot = 0
for r in d['roofline_shapes']:
    tot += r.get('ms_per_step', 0) or 0
for r in d['roofline_shapes'][:16]:
    print(f"   {r['kernel']:34s} wg={r['workgroups']:5d} n={r.get('launches_per_step')} us={r['avg_launch_us']:7.2f} ms/step={r.get('ms_per_step')} frac={r['frac']}")
PY
done
# the pre-change dispatch for comparison: fused winblk / ctblk off in the x2 mode only is not a knob; run f16 for the box's reference speed
timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-secondary --no-modes --prof-steps 0 > gpurun_out/r3c19_f16.json 2>> gpurun_out/r3c19.err
python -c "
import json; d = json.load(open('gpurun_out/r3c19_f16.json')); print('f16', d['ms_per_step'], d['value'], d['parity']['logits_max_abs_err'])"
tail -5 gpurun_out/r3c19.err
