#!/bin/bash
# r06 call 16: the two steps in flight partitioned in SPACE: CU-masked streams (hipExtStreamCreateWithCUMask), graph replay and eager
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r6c16_summary.log
: > $S
ab() {
  E="$1"; shift
  env $E timeout 400 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-secondary --no-modes --no-train-step --prof-steps 0 "$@" > gpurun_out/r6c16_ab.json 2>> gpurun_out/r6c16_ab.err
  python - "$E $*" <<'PY' >> gpurun_out/r6c16_summary.log
import json, sys
try:
    d = json.loads(open('gpurun_out/r6c16_ab.json').read().strip().splitlines()[-1])
    print(f"{sys.argv[1][:80]:80s} {d['ms_per_step']:.3f} ms/step {d['value']:.1f} img/s")
except Exception as e:
    print(sys.argv[1][:80], "FAILED", e)
PY
}
for round in 1 2; do
  ab X=1
  ab X=1 --cu-mask halves
  ab X=1 --cu-mask interleaved
  ab X=1 --cu-mask xcd
done
tail -12 gpurun_out/r6c16_ab.err >> $S
cat $S | cut -c1-330
