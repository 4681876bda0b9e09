#!/bin/bash
# r06 call 9: whole-batch steps in flight (PipelinedInference): --inflight 1 / 2 / 3, three interleaved rounds, 50 steps each
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r6c9_summary.log
: > $S
ab() {
  timeout 400 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary --no-modes --no-train-step --prof-steps 0 "$@" > gpurun_out/r6c9_ab.json 2>> gpurun_out/r6c9_ab.err
  python - "$*" <<'PY' >> gpurun_out/r6c9_summary.log
import json, sys
try:
    d = json.loads(open('gpurun_out/r6c9_ab.json').read().strip().splitlines()[-1])
    print(f"{sys.argv[1][:50]:50s} {d['ms_per_step']:.3f} ms/step {d['value']:.1f} img/s err {d['parity']['logits_max_abs_err'] if d.get('parity') else None} ({(d.get('parity') or {}).get('images')} img)")
except Exception as e:
    print(sys.argv[1][:50], "FAILED", e)
PY
}
for round in 1 2 3; do
  ab --inflight 1
  ab --inflight 2
  ab --inflight 3
done
ab --inflight 2 --streams 1 --join-from 0
ab --inflight 3 --streams 1 --join-from 0
ab --inflight 2 --join-from 0
tail -5 gpurun_out/r6c9_ab.err >> $S
cat $S | cut -c1-330
