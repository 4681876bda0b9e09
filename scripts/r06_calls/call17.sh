#!/bin/bash
# r06 call 17: winmlp<256> as FOUR waves x 128 rows (one workgroup per CU, one wave per SIMD, up to 512 registers): ring 2 / 4 deep, pipelined loop; kernel test, launch time alone, step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r6c17_summary.log
: > $S
for f in 4 5 6; do
  FVIT_TUNE_win_mlp256=$f timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "mlp" > gpurun_out/r6c17_tests.log 2>&1
  echo "tests win_mlp256=$f rc=$?" >> $S; tail -2 gpurun_out/r6c17_tests.log | cut -c1-200 >> $S
done
ab() {
  E="$1"; shift
  env $E timeout 400 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-secondary --no-modes --no-train-step --prof-steps 1 "$@" > gpurun_out/r6c17_ab.json 2>> gpurun_out/r6c17_ab.err
  python - "$E $*" <<'PY' >> gpurun_out/r6c17_summary.log
import json, sys
try:
    d = json.loads(open('gpurun_out/r6c17_ab.json').read().strip().splitlines()[-1])
    dd = json.load(open('gpurun_out/bench_detail.json'))
    ks = {r['kernel'] + 'x' + str(r['workgroups']): r['avg_launch_us'] for r in dd.get('roofline_shapes', [])}
    pick = ' '.join(f"{k[:24]}={v}" for k, v in ks.items() if k.startswith(('winmlp_kernel<256',)))
    print(f"{sys.argv[1][:40]:40s} {d['ms_per_step']:.3f} ms/step {d['value']:.1f} img/s err {d['parity']['logits_max_abs_err'] if d.get('parity') else None} | {pick}")
except Exception as e:
    print(sys.argv[1][:40], "FAILED", e)
PY
}
for round in 1 2; do
  ab FVIT_TUNE_win_mlp256=2
  ab FVIT_TUNE_win_mlp256=4
  ab FVIT_TUNE_win_mlp256=5
  ab FVIT_TUNE_win_mlp256=6
  ab FVIT_TUNE_win_mlp256=1
done
tail -3 gpurun_out/r6c17_ab.err >> $S
cat $S | cut -c1-330
