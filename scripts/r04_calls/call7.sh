#!/bin/bash
# r04 call 7: knob sweep in the new launch structure (2 stream shards, stage 3 joined and alone on the chip): options that were negative under three
# concurrent shards because they widen a kernel (sibling splits, the persistent one-launch stage 3, 8-wave MLP) are re-measured
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
ab() {
  E=$1; shift
  env $E timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary --no-modes --prof-steps 1 "$@" > gpurun_out/r4c7_ab.json 2>> gpurun_out/r4c7_ab.err
  python - "$E $*" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/r4c7_ab.json').read().strip().splitlines()[-1])
r = d.get('roofline') or {}
print(f"{sys.argv[1][:60]:60s} {d['ms_per_step']:.4f} ms/step {d['value']:.0f} img/s err {d['parity']['logits_max_abs_err']} dom {r.get('kernel')} {r.get('frac')}")
PY
}
ab X=0
ab FVIT_TUNE_win_stage3=1
ab FVIT_TUNE_win_mlp_split=2
ab FVIT_TUNE_win_blk_split=2
ab FVIT_TUNE_win_mlp256=1
ab FVIT_TUNE_win_mlp256=3
ab FVIT_TUNE_ab_variant=1
ab FVIT_TUNE_ab_stagger=1
ab X=0
ab FVIT_TUNE_ct_variant=0
ab FVIT_TUNE_conv_halo_grid=256
ab FVIT_TUNE_stem_fused_grid=256
ab X=0 --shard-sizes 120,136
ab X=0 --shard-sizes 136,120
tail -3 gpurun_out/r4c7_ab.err
