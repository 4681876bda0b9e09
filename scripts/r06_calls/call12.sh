#!/bin/bash
# r06 call 12: (a) winmlp<256> 128-row form in the whole-batch / 2-in-flight regime; (b) the secondary configs with steps in flight
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r6c12_summary.log
: > $S
ab() {
  E="$1"; shift
  env $E timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-modes --no-train-step --prof-steps 0 "$@" > gpurun_out/r6c12_ab.json 2>> gpurun_out/r6c12_ab.err
  python - "$E $*" <<'PY' >> gpurun_out/r6c12_summary.log
import json, sys
try:
    d = json.loads(open('gpurun_out/r6c12_ab.json').read().strip().splitlines()[-1])
    print(f"{sys.argv[1][:140]:140s} {d['ms_per_step']:.3f} ms/step {d['value']:.1f} img/s")
except Exception as e:
    print(sys.argv[1][:140], "FAILED", e)
PY
}
B="--steps 50 --warmup 10 --inflight 2 --streams 1 --join-from 0"
for round in 1 2 3; do
  ab X=1 $B
  ab FVIT_TUNE_win_mlp256=1 $B
done
F4="--model faster_vit_4_224 --batch 128 --steps 12 --warmup 3"
for round in 1 2; do
  ab X=1 $F4 --streams 2 --join-from 0 --operand f16x3 --precise
  ab X=1 $F4 --streams 2 --join-from 0 --operand f16x3 --precise --inflight 2
  ab X=1 $F4 --streams 1 --join-from 0 --operand f16x3 --precise --inflight 2
  ab X=1 $F4 --streams 3 --join-from 0
  ab X=1 $F4 --streams 2 --join-from 0 --inflight 2
  ab X=1 $F4 --streams 1 --join-from 0 --inflight 2
  ab X=1 $F4 --streams 1 --join-from 0 --inflight 3
done
AR="--model faster_vit_4_any_res --batch 8 --input-size 576x960 --steps 12 --warmup 3"
KW="{'resolution':[576,960],'window_size':[7,7,12,6],'ct_size':2}"
for round in 1 2; do
  ab X=1 $AR --model-kwargs "$KW" --streams 2 --join-from 0 --operand f16x3 --precise
  ab X=1 $AR --model-kwargs "$KW" --streams 2 --join-from 0 --operand f16x3 --precise --inflight 2
  ab X=1 $AR --model-kwargs "$KW" --streams 1 --join-from 0 --operand f16x3 --precise --inflight 2
  ab X=1 $AR --model-kwargs "$KW" --streams 1 --join-from 0 --operand f16x3 --precise --inflight 3
  ab X=1 $AR --model-kwargs "$KW" --streams 2 --join-from 0
  ab X=1 $AR --model-kwargs "$KW" --streams 1 --join-from 0 --inflight 2
  ab X=1 $AR --model-kwargs "$KW" --streams 1 --join-from 0 --inflight 3
done
tail -5 gpurun_out/r6c12_ab.err >> $S
cat $S | cut -c1-500
