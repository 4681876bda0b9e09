#!/bin/bash
# r04 call 11: "precise deploy" = 16-bit maps, two-term weights in every implicit-GEMM conv (FVIT_CONV_WEIGHT_TERMS=2) + x3 HAT stages: error and images/s
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
ab() {
  E=$1; shift
  env $E timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-modes --prof-steps 1 "$@" > gpurun_out/r4c11_ab.json 2>> gpurun_out/r4c11_ab.err
  python - "$E $*" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/r4c11_ab.json').read().strip().splitlines()[-1])
print(f"{sys.argv[1][:120]:120s} {d['ms_per_step']:.3f} ms/step {d['value']:.1f} img/s err {d['parity']['logits_max_abs_err'] if d.get('parity') else None} on {d['parity']['logits_abs_max'] if d.get('parity') else None}")
PY
}
F4="--model faster_vit_4_224 --batch 128 --streams 3 --join-from 0"
AR="--model faster_vit_4_any_res --batch 8 --input-size 576x960 --streams 2 --join-from 0"
KW="{'resolution':[576,960],'window_size':[7,7,12,6],'ct_size':2}"
ab FVIT_CONV_WEIGHT_TERMS=2 $F4 --operand f16x3
ab FVIT_CONV_WEIGHT_TERMS=2 $F4 --operand f16x2
ab FVIT_CONV_WEIGHT_TERMS=1 $F4 --operand f16x3
ab FVIT_CONV_WEIGHT_TERMS=2 $AR --model-kwargs "$KW" --operand f16x3
ab FVIT_CONV_WEIGHT_TERMS=2 --operand f16x3
ab FVIT_CONV_WEIGHT_TERMS=2 --operand f16x2
ab FVIT_CONV_WEIGHT_TERMS=2 --operand f16
tail -3 gpurun_out/r4c11_ab.err
