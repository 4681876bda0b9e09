#!/bin/bash
# r06 call 7: attnblk2_kernel (wave per (window, head)): kernel tests, micro-benchmark vs attnblk_kernel, step A/B ab_variant 0 / 3
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r6c7_summary.log
: > $S
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "attn_block" > gpurun_out/r6c7_tests.log 2>&1
echo "tests rc=$?" >> $S; tail -6 gpurun_out/r6c7_tests.log | cut -c1-300 >> $S
timeout 300 python scripts/bench_attnblk.py 53 512 a0v0,a0v3 >> $S 2>&1
timeout 300 python scripts/bench_attnblk.py 53 1024 a0v0,a0v3 >> $S 2>&1
ab() {
  E=$1; shift
  env $E timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-modes --no-train-step --prof-steps 1 "$@" > gpurun_out/r6c7_ab.json 2>> gpurun_out/r6c7_ab.err
  python - "$E $*" <<'PY' >> gpurun_out/r6c7_summary.log
import json, sys
try:
    d = json.loads(open('gpurun_out/r6c7_ab.json').read().strip().splitlines()[-1])
    dd = json.load(open('gpurun_out/bench_detail.json'))
    ks = {r['kernel'] + 'x' + str(r['workgroups']): r['avg_launch_us'] for r in dd.get('roofline_shapes', [])}
    pick = ' '.join(f"{k[:18]}={v}" for k, v in ks.items() if k.startswith(('attnblk', 'winmlp')))
    print(f"{sys.argv[1][:40]:40s} {d['ms_per_step']:.3f} ms/step {d['value']:.1f} img/s err {d['parity']['logits_max_abs_err'] if d.get('parity') else None} ({(d.get('parity') or {}).get('images')} img) | {pick}")
except Exception as e:
    print(sys.argv[1][:40], "FAILED", e)
PY
}
for round in 1 2 3; do
  ab FVIT_TUNE_ab_variant=0
  ab FVIT_TUNE_ab_variant=3
done
tail -3 gpurun_out/r6c7_ab.err >> $S
cat $S | cut -c1-330
