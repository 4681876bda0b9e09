#!/bin/bash
# r06 call 20: the remaining small knobs in the new launch structure (headline)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r6c20_summary.log
: > $S
ab() {
  E="$1"; shift
  env $E timeout 400 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-secondary --no-modes --no-train-step --prof-steps 0 "$@" > gpurun_out/r6c20_ab.json 2>> gpurun_out/r6c20_ab.err
  python - "$E $*" <<'PY' >> gpurun_out/r6c20_summary.log
import json, sys
try:
    d = json.loads(open('gpurun_out/r6c20_ab.json').read().strip().splitlines()[-1])
    print(f"{sys.argv[1][:70]:70s} {d['ms_per_step']:.3f} ms/step {d['value']:.1f} img/s")
except Exception as e:
    print(sys.argv[1][:70], "FAILED", e)
PY
}
for round in 1 2; do
  ab X=1
  ab FVIT_TUNE_ab_stagger=1
  ab FVIT_TUNE_ct8_depth=2
  ab FVIT_TUNE_ct8_depth=4
  ab FVIT_TUNE_conv64_variant=1
  ab FVIT_TUNE_conv128_narrow=1
  ab FVIT_TUNE_conv_halo_grid=768
  ab FVIT_TUNE_stem_fused_grid=768
  ab FVIT_TUNE_mlp_stagger=1
done
tail -3 gpurun_out/r6c20_ab.err >> $S
cat $S | cut -c1-200
