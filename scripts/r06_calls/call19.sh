#!/bin/bash
# r06 call 19: gemm256_min_tiles (the 256 x 256 ping-pong tile's minimum tile count) in the new launch structure: FasterViT-4 16-bit / precise, any-res precise
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r6c19_summary.log
: > $S
ab() {
  E="$1"; shift
  env $E timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-modes --no-train-step --prof-steps 0 "$@" > gpurun_out/r6c19_ab.json 2>> gpurun_out/r6c19_ab.err
  python - "$E $*" <<'PY' >> gpurun_out/r6c19_summary.log
import json, sys
try:
    d = json.loads(open('gpurun_out/r6c19_ab.json').read().strip().splitlines()[-1])
    print(f"{sys.argv[1][:60]:60s} {sys.argv[1][-40:]:40s} {d['ms_per_step']:.3f} ms/step {d['value']:.1f} img/s")
except Exception as e:
    print(sys.argv[1][:150], "FAILED", e)
PY
}
F4="--model faster_vit_4_224 --batch 128 --steps 12 --warmup 3 --streams 1 --join-from 0 --inflight 2"
AR="--model faster_vit_4_any_res --batch 8 --input-size 576x960 --steps 12 --warmup 3 --streams 1 --join-from 0 --inflight 3"
KW="{'resolution':[576,960],'window_size':[7,7,12,6],'ct_size':2}"
for round in 1 2; do
for t in 192 128 96 64 32; do
  ab FVIT_TUNE_gemm256_min_tiles=$t $F4
  ab FVIT_TUNE_gemm256_min_tiles=$t $F4 --operand f16x3 --precise
  ab FVIT_TUNE_gemm256_min_tiles=$t $AR --model-kwargs "$KW" --operand f16x3 --precise
  ab FVIT_TUNE_gemm256_min_tiles=$t $AR --model-kwargs "$KW"
done
done
tail -5 gpurun_out/r6c19_ab.err >> $S
cat $S | cut -c1-400
