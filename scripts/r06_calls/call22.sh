#!/bin/bash
# r06 call 22: dense-K implicit-GEMM conv (fvit_conv3x3_nhwc_dense: the pad channels of the input map leave the contraction; FasterViT-4: 196 of 256, 392 of 448):
# kernel tests, the model-level parity tests that run the deploy plans, then A/B FVIT_CONV_DENSE_K=0|1 on both secondary configurations in both plans
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r6c22_summary.log
: > $S
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_px.py -q -m gpu -k "conv3x3" -x 2>&1 | tail -5 >> $S
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_px.py tests/test_gpu_precision_modes.py tests/test_gpu_determinism.py -q -m gpu -x 2>&1 | tail -5 >> $S
ab() {
  E="$1"; shift
  env $E timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-modes --no-train-step --prof-steps 0 "$@" > gpurun_out/r6c22_ab.json 2>> gpurun_out/r6c22_ab.err
  python - "$E $*" <<'PY' >> gpurun_out/r6c22_summary.log
import json, sys
try:
    d = json.loads(open('gpurun_out/r6c22_ab.json').read().strip().splitlines()[-1])
    par = d.get('parity') or {}
    print(f"{sys.argv[1][:130]:130s} {d['ms_per_step']:.3f} ms/step {d['value']:.1f} img/s err {par.get('logits_max_abs_err')} ({par.get('images')} img)")
except Exception as e:
    print(sys.argv[1][:130], "FAILED", e)
PY
}
F4="--model faster_vit_4_224 --batch 128 --steps 12 --warmup 3 --streams 1 --join-from 0 --inflight 2"
AR="--model faster_vit_4_any_res --batch 8 --input-size 576x960 --steps 12 --warmup 3 --streams 1 --join-from 0 --inflight 3"
KW="{'resolution':[576,960],'window_size':[7,7,12,6],'ct_size':2}"
for rep in 1 2; do
  for k in FVIT_CONV_DENSE_K=0 FVIT_CONV_DENSE_K=1; do
    ab $k $F4
    ab $k $F4 --operand f16x3 --precise
  done
done
for k in FVIT_CONV_DENSE_K=0 FVIT_CONV_DENSE_K=1 FVIT_CONV_DENSE_K=0 FVIT_CONV_DENSE_K=1; do
  ab $k $AR --model-kwargs "$KW" --operand f16x3 --precise
done
for k in FVIT_CONV_DENSE_K=0 FVIT_CONV_DENSE_K=1; do ab $k $AR --model-kwargs "$KW"; done
tail -5 gpurun_out/r6c22_ab.err >> $S
cat $S | cut -c1-300
