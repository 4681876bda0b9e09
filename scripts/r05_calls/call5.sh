#!/bin/bash
# r05 call 5: deterministic split-K of the small-grid residual GEMMs (carrier-token branch: 56 workgroups x 98-147 K steps in the x3 modes) inside the precise plan
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r5c5_summary.log
: > $S
ab() {
  E=$1; shift
  env $E timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-modes --no-train-step --prof-steps 1 "$@" > gpurun_out/r5c5_ab.json 2>> gpurun_out/r5c5_ab.err
  python - "$E $*" <<'PY' >> gpurun_out/r5c5_summary.log
import json, sys
try:
    d = json.loads(open('gpurun_out/r5c5_ab.json').read().strip().splitlines()[-1])
    print(f"{sys.argv[1][:110]:110s} {d['ms_per_step']:.3f} ms/step {d['value']:.1f} img/s err {d['parity']['logits_max_abs_err'] if d.get('parity') else None}")
except Exception as e:
    print(sys.argv[1][:110], "FAILED", e)
PY
}
F4="--model faster_vit_4_224 --batch 128 --join-from 0 --operand f16x3 --precise --streams 2"
AR="--model faster_vit_4_any_res --batch 8 --input-size 576x960 --join-from 0 --operand f16x3 --precise --streams 2"
KW="{'resolution':[576,960],'window_size':[7,7,12,6],'ct_size':2}"
for i in 1 2; do
ab FVIT_TUNE_gemm_splitk=0 $F4
ab FVIT_TUNE_gemm_splitk=1 $F4
done
ab FVIT_TUNE_gemm_splitk=0 $AR --model-kwargs "$KW"
ab FVIT_TUNE_gemm_splitk=1 $AR --model-kwargs "$KW"
ab FVIT_TUNE_gemm_splitk=0 $F4 --streams 1
ab FVIT_TUNE_gemm_splitk=1 $F4 --streams 1
tail -3 gpurun_out/r5c5_ab.err >> $S
cat $S | cut -c1-300
