#!/bin/bash
# r06 call 11: whole-batch launches + 2 steps in flight, attention-block variants (library default back to ab_variant 0) and grid knobs, three rounds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r6c11_summary.log
: > $S
ab() {
  E="$1"; shift
  env $E timeout 400 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary --no-modes --no-train-step --prof-steps 0 "$@" > gpurun_out/r6c11_ab.json 2>> gpurun_out/r6c11_ab.err
  python - "$E $*" <<'PY' >> gpurun_out/r6c11_summary.log
import json, sys
try:
    d = json.loads(open('gpurun_out/r6c11_ab.json').read().strip().splitlines()[-1])
    print(f"{sys.argv[1][:100]:100s} {d['ms_per_step']:.3f} ms/step {d['value']:.1f} img/s")
except Exception as e:
    print(sys.argv[1][:100], "FAILED", e)
PY
}
B="--inflight 2 --streams 1 --join-from 0"
for round in 1 2 3; do
  ab X=1 --inflight 1
  ab FVIT_TUNE_ab_variant=0 $B
  ab FVIT_TUNE_ab_variant=1 $B
  ab FVIT_TUNE_ab_variant=2 $B
  ab "FVIT_TUNE_ab_variant=2 FVIT_TUNE_stem_fused_grid=256" $B
  ab "FVIT_TUNE_ab_variant=2 FVIT_TUNE_ct_variant=2" $B
  ab "FVIT_TUNE_ab_variant=2 FVIT_TUNE_win_blk_split=2" $B
  ab "FVIT_TUNE_ab_variant=2 FVIT_TUNE_win_mlp_split=2" $B
  ab FVIT_TUNE_ab_variant=2 --inflight 3 --streams 1 --join-from 0
done
tail -5 gpurun_out/r6c11_ab.err >> $S
cat $S | cut -c1-400
