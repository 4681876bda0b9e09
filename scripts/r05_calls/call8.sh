#!/bin/bash
# r05 call 8: launch-structure knobs of the precise FasterViT-4 plan (stage-3 join, 256 x 256 tile threshold)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r5c8_summary.log
: > $S
ab() {
  E=$1; shift
  env $E timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-modes --no-train-step --prof-steps 0 "$@" > gpurun_out/r5c8_ab.json 2>> gpurun_out/r5c8_ab.err
  python - "$E $*" <<'PY' >> gpurun_out/r5c8_summary.log
import json, sys
try:
    d = json.loads(open('gpurun_out/r5c8_ab.json').read().strip().splitlines()[-1])
    print(f"{sys.argv[1][:150]:150s} {d['ms_per_step']:.3f} ms/step {d['value']:.1f} img/s")
except Exception as e:
    print(sys.argv[1][:150], "FAILED", e)
PY
}
F4="--model faster_vit_4_224 --batch 128 --operand f16x3 --precise --streams 2"
ab X=1 $F4 --join-from 0
ab X=1 $F4 --join-from 3
ab FVIT_TUNE_gemm256_min_tiles=150 $F4 --join-from 3
ab FVIT_TUNE_gemm256_min_tiles=90 $F4 --join-from 0
ab X=1 $F4 --join-from 2
ab X=1 $F4 --join-from 0
AR="--model faster_vit_4_any_res --batch 8 --input-size 576x960 --operand f16x3 --precise --streams 2"
KW="{'resolution':[576,960],'window_size':[7,7,12,6],'ct_size':2}"
ab X=1 $AR --model-kwargs "$KW" --join-from 0
ab X=1 $AR --model-kwargs "$KW" --join-from 3
ab FVIT_TUNE_gemm256_min_tiles=90 $AR --model-kwargs "$KW" --join-from 0
tail -3 gpurun_out/r5c8_ab.err >> $S
cat $S | cut -c1-300
