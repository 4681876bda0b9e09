#!/bin/bash
# r06 call 24: gemm_pp_kernel with a 3-deep ACTIVATION ring (fvit_tune gemm_pp_xring = 3: X(t + 2) requested during K tile t, 160 KiB of LDS, vmcnt(8)):
# the GEMM kernel tests under the knob (the 256 x 256 tile must stay bitwise the 128 x 128 tile's result), then A/B on FasterViT-4 in both plans
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r6c24_summary.log
: > $S
FVIT_TUNE_gemm_pp_xring=3 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_x3.py -q -m gpu -k "gemm" -x 2>&1 | tail -3 >> $S
ab() {
  E="$1"; shift
  env $E timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-modes --no-train-step --prof-steps 0 "$@" > gpurun_out/r6c24_ab.json 2>> gpurun_out/r6c24_ab.err
  python - "$E $*" <<'PY' >> gpurun_out/r6c24_summary.log
import json, sys
try:
    d = json.loads(open('gpurun_out/r6c24_ab.json').read().strip().splitlines()[-1])
    par = d.get('parity') or {}
    print(f"{sys.argv[1][:120]:120s} {d['ms_per_step']:.3f} ms/step {d['value']:.1f} img/s err {par.get('logits_max_abs_err')} ({par.get('images')} img)")
except Exception as e:
    print(sys.argv[1][:120], "FAILED", e)
PY
}
F4="--model faster_vit_4_224 --batch 128 --steps 12 --warmup 3 --streams 1 --join-from 0 --inflight 2"
for rep in 1 2; do
  for k in FVIT_TUNE_gemm_pp_xring=2 FVIT_TUNE_gemm_pp_xring=3; do
    ab $k $F4
    ab $k $F4 --operand f16x3 --precise
  done
done
tail -5 gpurun_out/r6c24_ab.err >> $S
cat $S | cut -c1-300
