#!/bin/bash
# r06 call 14: structure knobs re-measured in the new regime (2 whole-batch steps in flight): fused vs GEMM-chain forms of stage 3 / stage 2, conv kernel choices
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r6c14_summary.log
: > $S
ab() {
  E="$1"; shift
  env $E timeout 400 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-secondary --no-modes --no-train-step --prof-steps 0 "$@" > gpurun_out/r6c14_ab.json 2>> gpurun_out/r6c14_ab.err
  python - "$E $*" <<'PY' >> gpurun_out/r6c14_summary.log
import json, sys
try:
    d = json.loads(open('gpurun_out/r6c14_ab.json').read().strip().splitlines()[-1])
    print(f"{sys.argv[1][:100]:100s} {d['ms_per_step']:.3f} ms/step {d['value']:.1f} img/s")
except Exception as e:
    print(sys.argv[1][:100], "FAILED", e)
PY
}
for round in 1 2; do
  ab X=1
  ab FVIT_TUNE_win_mlp=0
  ab FVIT_TUNE_win_fused=0
  ab "FVIT_TUNE_win_mlp=0 FVIT_TUNE_win_fused=0"
  ab FVIT_TUNE_win_mlp256=0
  ab FVIT_TUNE_attn_fused=0
  ab FVIT_TUNE_ct_fused=0
  ab FVIT_TUNE_conv_band=0
  ab FVIT_TUNE_conv_halo=0
  ab FVIT_TUNE_win_mlp_pipe=0
  ab X=1 --inflight 2 --streams 2 --join-from 3
done
tail -5 gpurun_out/r6c14_ab.err >> $S
cat $S | cut -c1-400
