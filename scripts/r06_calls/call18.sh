#!/bin/bash
# r06 call 18: FasterViT-4 (16-bit plan and precise plan) in the new launch structure (whole-batch launches, 2 steps in flight): GEMM / conv knobs that were tuned for shard-sized launches
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r6c18_summary.log
: > $S
ab() {
  E="$1"; shift
  env $E timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-modes --no-train-step --prof-steps 0 "$@" > gpurun_out/r6c18_ab.json 2>> gpurun_out/r6c18_ab.err
  python - "$E $*" <<'PY' >> gpurun_out/r6c18_summary.log
import json, sys
try:
    d = json.loads(open('gpurun_out/r6c18_ab.json').read().strip().splitlines()[-1])
    print(f"{sys.argv[1][:150]:150s} {d['ms_per_step']:.3f} ms/step {d['value']:.1f} img/s")
except Exception as e:
    print(sys.argv[1][:150], "FAILED", e)
PY
}
F4="--model faster_vit_4_224 --batch 128 --steps 12 --warmup 3 --streams 1 --join-from 0 --inflight 2"
for k in X=1 FVIT_TUNE_gemm_pp=0 FVIT_TUNE_gemm256_min_tiles=96 FVIT_TUNE_gemm256_min_tiles=400 FVIT_TUNE_gemm_bm64_max_grid=0 FVIT_TUNE_gemm_bm64_max_grid=800 FVIT_TUNE_ln_gemm=1 FVIT_TUNE_gemm_splitk=1 FVIT_TUNE_conv_n128_ragged=0 FVIT_TUNE_gemm_nw8_max_grid=600 X=2; do
  ab $k $F4
done
for k in X=1 FVIT_TUNE_gemm_pp=0 FVIT_TUNE_gemm256_min_tiles=96 FVIT_TUNE_gemm256_min_tiles=400 FVIT_TUNE_gemm_bm64_max_grid=0 FVIT_TUNE_gemm_splitk=1 FVIT_TUNE_gemm_x3_dual=0 X=2; do
  ab $k $F4 --operand f16x3 --precise
done
tail -5 gpurun_out/r6c18_ab.err >> $S
cat $S | cut -c1-400
