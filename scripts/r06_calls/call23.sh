#!/bin/bash
# r06 call 23: conv3x3_kernel with a 3-deep ACTIVATION ring (XR = 3: X(kt + 2) requested while step kt runs, counted vmcnt; 80 KiB = still 2 workgroups per CU):
# conv kernel tests on the new default, then A/B FVIT_TUNE_conv_xring=2|3 on the headline and on FasterViT-4 (both plans), interleaved
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r6c23_summary.log
: > $S
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_px.py -q -m gpu -k "conv3x3" -x 2>&1 | tail -3 >> $S
ab() {
  E="$1"; shift
  env $E timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-modes --no-train-step --prof-steps 0 "$@" > gpurun_out/r6c23_ab.json 2>> gpurun_out/r6c23_ab.err
  python - "$E $*" <<'PY' >> gpurun_out/r6c23_summary.log
import json, sys
try:
    d = json.loads(open('gpurun_out/r6c23_ab.json').read().strip().splitlines()[-1])
    par = d.get('parity') or {}
    print(f"{sys.argv[1][:120]:120s} {d['ms_per_step']:.3f} ms/step {d['value']:.1f} img/s err {par.get('logits_max_abs_err')} ({par.get('images')} img)")
except Exception as e:
    print(sys.argv[1][:120], "FAILED", e)
PY
}
F4="--model faster_vit_4_224 --batch 128 --steps 12 --warmup 3 --streams 1 --join-from 0 --inflight 2"
for rep in 1 2; do
  for k in FVIT_TUNE_conv_xring=2 FVIT_TUNE_conv_xring=3; do
    ab $k --steps 50 --warmup 10
    ab $k $F4
  done
done
for k in FVIT_TUNE_conv_xring=2 FVIT_TUNE_conv_xring=3; do
  ab $k $F4 --operand f16x3 --precise
done
tail -5 gpurun_out/r6c23_ab.err >> $S
cat $S | cut -c1-300
