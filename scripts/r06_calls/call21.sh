#!/bin/bash
# r06 call 21: stage 3 as ONE launch of persistent per-window workgroups (fvit_stage3.hip, win_stage3 = 1) in the new launch structure; parity once
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r6c21_summary.log
: > $S
ab() {
  E="$1"; shift
  env $E timeout 400 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-secondary --no-modes --no-train-step --prof-steps 1 "$@" > gpurun_out/r6c21_ab.json 2>> gpurun_out/r6c21_ab.err
  python - "$E $*" <<'PY' >> gpurun_out/r6c21_summary.log
import json, sys
try:
    d = json.loads(open('gpurun_out/r6c21_ab.json').read().strip().splitlines()[-1])
    dd = json.load(open('gpurun_out/bench_detail.json'))
    ks = {r['kernel'] + 'x' + str(r['workgroups']): (r['avg_launch_us'], r['launches_per_step']) for r in dd.get('roofline_shapes', [])}
    pick = ' '.join(f"{k[:26]}={v}" for k, v in ks.items() if k.startswith(('win_stage3', 'winblk', 'winmlp_kernel<512')))
    print(f"{sys.argv[1][:40]:40s} {d['ms_per_step']:.3f} ms/step {d['value']:.1f} img/s err {d['parity']['logits_max_abs_err'] if d.get('parity') else None} | {pick}")
except Exception as e:
    print(sys.argv[1][:40], "FAILED", e)
PY
}
for round in 1 2 3; do
  ab FVIT_TUNE_win_stage3=0
  ab FVIT_TUNE_win_stage3=1
done
tail -3 gpurun_out/r6c21_ab.err >> $S
cat $S | cut -c1-330
