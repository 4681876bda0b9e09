#!/bin/bash
# r06 call 6: the contiguous-run gather of the fused stem (fp32 channels-last input): kernel + parity tests, then A/B stem_nhwc3 0 / 1, three interleaved pairs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r6c6_summary.log
: > $S
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_x3.py -q -m gpu -k "stem or fvit0 or logits or long_windows" > gpurun_out/r6c6_tests.log 2>&1
echo "tests rc=$?" >> $S; tail -6 gpurun_out/r6c6_tests.log | cut -c1-300 >> $S
ab() {
  E=$1; shift
  env $E timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-modes --no-train-step --prof-steps 1 "$@" > gpurun_out/r6c6_ab.json 2>> gpurun_out/r6c6_ab.err
  python - "$E $*" <<'PY' >> gpurun_out/r6c6_summary.log
import json, sys
try:
    d = json.loads(open('gpurun_out/r6c6_ab.json').read().strip().splitlines()[-1])
    dd = json.load(open('gpurun_out/bench_detail.json'))
    ks = {r['kernel'] + 'x' + str(r['workgroups']): r['avg_launch_us'] for r in dd.get('roofline_shapes', [])}
    pick = ' '.join(f"{k[:16]}={v}" for k, v in ks.items() if k.startswith(('stem', 'conv3x3_c64')))
    print(f"{sys.argv[1][:40]:40s} {d['ms_per_step']:.3f} ms/step {d['value']:.1f} img/s err {d['parity']['logits_max_abs_err'] if d.get('parity') else None} ({(d.get('parity') or {}).get('images')} img) | {pick}")
except Exception as e:
    print(sys.argv[1][:40], "FAILED", e)
PY
}
for round in 1 2 3; do
  ab FVIT_TUNE_stem_nhwc3=0
  ab FVIT_TUNE_stem_nhwc3=1
done
tail -3 gpurun_out/r6c6_ab.err >> $S
cat $S | cut -c1-330
