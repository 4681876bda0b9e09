#!/bin/bash
# r06 call 10: whole-batch launches + 2 steps in flight: knob sweep in that regime (kernel variants tuned for shard-sized launches may flip), parity once
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r6c10_summary.log
: > $S
ab() {
  E="$1"; shift
  env $E timeout 400 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary --no-modes --no-train-step --prof-steps 0 "$@" > gpurun_out/r6c10_ab.json 2>> gpurun_out/r6c10_ab.err
  python - "$E $*" <<'PY' >> gpurun_out/r6c10_summary.log
import json, sys
try:
    d = json.loads(open('gpurun_out/r6c10_ab.json').read().strip().splitlines()[-1])
    print(f"{sys.argv[1][:90]:90s} {d['ms_per_step']:.3f} ms/step {d['value']:.1f} img/s")
except Exception as e:
    print(sys.argv[1][:90], "FAILED", e)
PY
}
B="--inflight 2 --streams 1 --join-from 0"
for round in 1 2; do
  ab X=1 --inflight 1
  ab X=1 $B
  ab FVIT_TUNE_ab_variant=1 $B
  ab FVIT_TUNE_ab_variant=2 $B
  ab "FVIT_TUNE_ab_variant=3 FVIT_TUNE_ab2_nwin=1" $B
  ab "FVIT_TUNE_ab_variant=3 FVIT_TUNE_ab2_nwin=2" $B
  ab FVIT_TUNE_win_mlp_pipe=0 $B
  ab FVIT_TUNE_conv_halo_grid=256 $B
  ab FVIT_TUNE_stem_fused_grid=256 $B
  ab X=1 --inflight 2 --streams 2 --join-from 2
  ab X=1 --inflight 4 --streams 1 --join-from 0
  ab X=1 --inflight 2 --streams 1 --join-from 0 --batch 128
done
# parity of the new default candidate (all 256 images) + kernel rows
timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary --no-modes --no-train-step --prof-steps 1 $B > gpurun_out/r6c10_parity.json 2>> gpurun_out/r6c10_ab.err
python - <<'PY' >> $S
import json
d = json.loads(open('gpurun_out/r6c10_parity.json').read().strip().splitlines()[-1])
print("parity run:", d['value'], d['ms_per_step'], d.get('parity'), d['config']['launch'])
PY
tail -5 gpurun_out/r6c10_ab.err >> $S
cat $S | cut -c1-400
