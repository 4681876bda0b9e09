#!/bin/bash
# r06 call 15: marginal cost of every kernel family in the NEW timed structure (2 whole-batch steps in flight): diagnosis build, one family's launches skipped at a time
# (fvit_tune ablate_skip bits: 1 winmlp<256>, 2 winmlp<512>, 4 winblk, 8 attnblk, 16 ctblk, 32 conv3x3 implicit GEMM + band, 64 halo conv, 128 fused stem; results are wrong by construction)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r6c15_summary.log
: > $S
ab() {
  E="$1"; shift
  env $E timeout 400 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-secondary --no-modes --no-train-step --prof-steps 0 "$@" > gpurun_out/r6c15_ab.json 2>> gpurun_out/r6c15_ab.err
  python - "$E $*" <<'PY' >> gpurun_out/r6c15_summary.log
import json, sys
try:
    d = json.loads(open('gpurun_out/r6c15_ab.json').read().strip().splitlines()[-1])
    print(f"{sys.argv[1][:80]:80s} {d['ms_per_step']:.3f} ms/step")
except Exception as e:
    print(sys.argv[1][:80], "FAILED", e)
PY
}
for round in 1 2; do
  for m in 0 1 8 16 25 2 4 6 32 64 128 224 31 255; do
    ab "FVIT_DIAG=1 FVIT_TUNE_ablate_skip=$m"
  done
done
tail -3 gpurun_out/r6c15_ab.err >> $S
cat $S | cut -c1-330
