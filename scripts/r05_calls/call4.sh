#!/bin/bash
# r05 call 4: 4-deep weight ring of the stage-2 MLP kernel (fvit_tune win_mlp256_depth = 4): same bits, A/B x 3 interleaved in one box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r5c4_summary.log
: > $S
timeout 600 python -m pytest tests/test_gpu_determinism.py -q -m gpu -k "order_stagger_knobs" > gpurun_out/r5c4_tests.log 2>&1
echo "knob tests rc=$?" >> $S; tail -3 gpurun_out/r5c4_tests.log | cut -c1-300 >> $S
ab() {
  E=$1; shift
  env $E timeout 400 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary --no-modes --no-train-step --prof-steps 1 "$@" > gpurun_out/r5c4_ab.json 2>> gpurun_out/r5c4_ab.err
  python - "$E $*" <<'PY' >> gpurun_out/r5c4_summary.log
import json, sys
try:
    d = json.loads(open('gpurun_out/r5c4_ab.json').read().strip().splitlines()[-1])
    r = d.get('roofline') or {}
    det = json.load(open('gpurun_out/bench_detail.json'))
    wm = [x for x in det.get('roofline_shapes', []) if x['kernel'].startswith('winmlp_kernel<256>')]
    print(f"{sys.argv[1][:60]:60s} {d['ms_per_step']:.4f} ms/step {d['value']:.0f} img/s err {d['parity']['logits_max_abs_err']} dom {r.get('kernel')} {r.get('avg_launch_us')} us frac {r.get('frac')} | winmlp<256> {wm[0]['avg_launch_us'] if wm else None} us")
except Exception as e:
    print(sys.argv[1][:60], "FAILED", e)
PY
}
for i in 1 2 3; do
ab FVIT_TUNE_win_mlp256_depth=2
ab FVIT_TUNE_win_mlp256_depth=4
done
tail -3 gpurun_out/r5c4_ab.err >> $S
cat $S | cut -c1-400
