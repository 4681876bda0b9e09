#!/bin/bash
# r06 call 2: marginal cost of every kernel family INSIDE the timed hipGraph (two stream shards + join): the diagnosis build with one family's launches
# skipped at a time (FVIT_TUNE_ablate_skip, results are wrong by construction; timing only), plus the per-block error measurement for the test bounds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r6c2_summary.log
: > $S
ab() {
  E=$1; shift
  env $E timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-modes --no-train-step --prof-steps 0 "$@" > gpurun_out/r6c2_ab.json 2>> gpurun_out/r6c2_ab.err
  python - "$E $*" <<'PY' >> gpurun_out/r6c2_summary.log
import json, sys
try:
    d = json.loads(open('gpurun_out/r6c2_ab.json').read().strip().splitlines()[-1])
    print(f"{sys.argv[1][:90]:90s} {d['ms_per_step']:.3f} ms/step {d['value']:.1f} img/s")
except Exception as e:
    print(sys.argv[1][:90], "FAILED", e)
PY
}
for round in 1 2; do
  ab "FVIT_DIAG=1 FVIT_TUNE_ct_nimg=1 FVIT_TUNE_ablate_skip=0"
  for bit in 1 2 4 8 16 32 64 128; do
    ab "FVIT_DIAG=1 FVIT_TUNE_ct_nimg=1 FVIT_TUNE_ablate_skip=$bit"
  done
  ab "FVIT_DIAG=1 FVIT_TUNE_ct_nimg=1 FVIT_TUNE_ablate_skip=25"
  ab "FVIT_DIAG=1 FVIT_TUNE_ct_nimg=1 FVIT_TUNE_ablate_skip=31"
  ab "FVIT_DIAG=1 FVIT_TUNE_ct_nimg=1 FVIT_TUNE_ablate_skip=224"
done
timeout 600 python scripts/measure_parity_margins.py > gpurun_out/r6c2_margins.log 2>&1
tail -12 gpurun_out/r6c2_margins.log >> $S
tail -3 gpurun_out/r6c2_ab.err >> $S
cat $S | cut -c1-200
