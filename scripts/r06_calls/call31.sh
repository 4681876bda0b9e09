#!/bin/bash
# r06 call 31: the PATCH form of the implicit-GEMM conv (conv3x3_kernel<.., HALO>, fvit_tune conv_patch): kernel tests (16-bit and two-term-map entry points), the conv alone
# on FasterViT-4's level-0 shape in both forms, then A/B FVIT_TUNE_conv_patch=0|1 on FasterViT-4 and any-res in both plans (parity from the bench's own oracle check)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r6c31_summary.log
: > $S
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_px.py -q -m gpu -s -k "patch_form" 2>&1 | grep -v "^$" | tail -14 | cut -c1-200 >> $S
for f in 0 1 0 1; do
  echo "== conv alone, 128 x 56 x 56 x 256 -> 256, conv_patch=$f" >> $S
  FVIT_DIAG=0 FVIT_TUNE_conv_patch=$f CONV_C=256 timeout 300 python scripts/bench_conv.py 128 56 56 gemm 2>&1 | grep "gemm " >> $S
done
ab() {
  E="$1"; shift
  env $E timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-modes --no-train-step --steps 12 --warmup 3 "$@" > gpurun_out/r6c31_ab.json 2>> gpurun_out/r6c31_ab.err
  python - "$E $*" <<'PY' >> gpurun_out/r6c31_summary.log
import json, sys
try:
    d = json.loads(open('gpurun_out/r6c31_ab.json').read().strip().splitlines()[-1])
    par = d.get('parity') or {}
    print(f"{sys.argv[1][:100]:100s} {d['ms_per_step']:.3f} ms/step {d['value']:.1f} img/s err {par.get('logits_max_abs_err')} ({par.get('images')} img)")
except Exception as e:
    print(sys.argv[1][:100], "FAILED", e)
PY
}
F4="--model faster_vit_4_224 --batch 128 --streams 1 --join-from 0 --inflight 2"
AR="--model faster_vit_4_any_res --batch 8 --input-size 576x960 --streams 1 --join-from 0 --inflight 3"
KW="{'resolution':[576,960],'window_size':[7,7,12,6],'ct_size':2}"
for rep in 1 2; do
  for k in FVIT_TUNE_conv_patch=0 FVIT_TUNE_conv_patch=1; do
    ab $k $F4
    ab $k $AR --model-kwargs "$KW"
  done
done
for k in FVIT_TUNE_conv_patch=0 FVIT_TUNE_conv_patch=1; do
  ab $k $F4 --operand f16x3 --precise
  ab $k $AR --model-kwargs "$KW" --operand f16x3 --precise
done
tail -3 gpurun_out/r6c31_ab.err >> $S
cat $S | cut -c1-250
