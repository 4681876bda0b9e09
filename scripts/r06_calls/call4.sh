#!/bin/bash
# r06 call 4: what bounds a winmlp super-chunk?  Diagnosis build, plain main loop (win_mlp_pipe=0), one part removed at a time (fvit_tune wm_ablate: 1 no weight
# loads in the loop, 2 no barrier, 4 GELU -> identity; results wrong by construction); per-launch event times of winmlp<256> / winmlp<512> and the step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r6c4_summary.log
: > $S
ab() {
  E=$1; shift
  env $E timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-modes --no-train-step --prof-steps 1 "$@" > gpurun_out/r6c4_ab.json 2>> gpurun_out/r6c4_ab.err
  python - "$E $*" <<'PY' >> gpurun_out/r6c4_summary.log
import json, sys
try:
    d = json.loads(open('gpurun_out/r6c4_ab.json').read().strip().splitlines()[-1])
    dd = json.load(open('gpurun_out/bench_detail.json'))
    ks = {r['kernel'] + 'x' + str(r['workgroups']): r['avg_launch_us'] for r in dd.get('roofline_shapes', [])}
    pick = ' '.join(f"{k.split('_kernel')[0]}{k.split('_kernel')[1][:5]}={v}" for k, v in ks.items() if k.startswith(('winmlp', 'ctblk', 'attnblk', 'winblk')))
    print(f"{sys.argv[1][:75]:75s} {d['ms_per_step']:.3f} ms/step | {pick}")
except Exception as e:
    print(sys.argv[1][:75], "FAILED", e)
PY
}
for pipe in 0 1; do
for a in 0 1 2 4 3 7; do
  ab "FVIT_DIAG=1 FVIT_TUNE_win_mlp_pipe=$pipe FVIT_TUNE_wm_ablate=$a"
done
done
tail -3 gpurun_out/r6c4_ab.err >> $S
cat $S | cut -c1-330
