#!/bin/bash
# r06 call 1: kernel / parity / determinism tests on the r06 kernels (interleaved GELU chains, carrier branch with two images per workgroup),
# then the headline A/B in ONE box, three interleaved rounds: r05 library vs r06 (ct_nimg 2) vs r06 with ct_nimg 1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r6c1_summary.log
: > $S
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_determinism.py -q -m gpu -x > gpurun_out/r6c1_tests.log 2>&1
echo "tests rc=$?" >> $S; tail -8 gpurun_out/r6c1_tests.log | cut -c1-300 >> $S
ab() {
  E=$1; shift
  env $E timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-modes --no-train-step --prof-steps 1 "$@" > gpurun_out/r6c1_ab.json 2>> gpurun_out/r6c1_ab.err
  python - "$E $*" <<'PY' >> gpurun_out/r6c1_summary.log
import json, sys
try:
    d = json.loads(open('gpurun_out/r6c1_ab.json').read().strip().splitlines()[-1])
    r = d.get('roofline') or {}
    print(f"{sys.argv[1][:90]:90s} {d['ms_per_step']:.3f} ms/step {d['value']:.1f} img/s err {d['parity']['logits_max_abs_err'] if d.get('parity') else None} dom {r.get('kernel')} {r.get('avg_launch_us')} us")
except Exception as e:
    print(sys.argv[1][:90], "FAILED", e)
PY
}
for round in 1 2 3; do
  ab FVIT_LIB_PATH=scripts/ab/libfvit_hip_r05.so
  ab FVIT_TUNE_ct_nimg=2
  ab FVIT_TUNE_ct_nimg=1
done
cp gpurun_out/bench_detail.json gpurun_out/r6c1_bench_detail_last.json 2>/dev/null
tail -3 gpurun_out/r6c1_ab.err >> $S
cat $S | cut -c1-300
