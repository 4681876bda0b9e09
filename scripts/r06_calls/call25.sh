#!/bin/bash
# r06 call 25: the shipped build reads every kernel's ablation mask through FVIT_ABL = compile-time 0 (no dead `if (p.ablate & ..)` splitting the hot loops of
# winmlp / attnblk / the halo conv / conv3x3 / mlp_fused).  A/B in one box against the previous commit's library (fastervit_amd/csrc/ab/libfvit_hip_prev.so,
# built from `git archive 53a81e3`, selected with FVIT_LIB_PATH); the logits must be bitwise the same (same arithmetic), so the parity figure must not move.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r6c25_summary.log
: > $S
ab() {
  E="$1"; shift
  env $E timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-modes --no-train-step --prof-steps 0 "$@" > gpurun_out/r6c25_ab.json 2>> gpurun_out/r6c25_ab.err
  python - "$E $*" <<'PY' >> gpurun_out/r6c25_summary.log
import json, sys
try:
    d = json.loads(open('gpurun_out/r6c25_ab.json').read().strip().splitlines()[-1])
    par = d.get('parity') or {}
    print(f"{sys.argv[1][:110]:110s} {d['ms_per_step']:.3f} ms/step {d['value']:.1f} img/s err {par.get('logits_max_abs_err')} ({par.get('images')} img, worst {par.get('worst_image')})")
except Exception as e:
    print(sys.argv[1][:110], "FAILED", e)
PY
}
PREV=FVIT_LIB_PATH=$GRAFT_REPO_ROOT/fastervit_amd/csrc/ab/libfvit_hip_prev.so
F4="--model faster_vit_4_224 --batch 128 --steps 12 --warmup 3 --streams 1 --join-from 0 --inflight 2"
for rep in 1 2 3; do
  ab $PREV --steps 50 --warmup 10
  ab X=new --steps 50 --warmup 10
done
ab $PREV $F4
ab X=new $F4
ab $PREV $F4
ab X=new $F4
# parity of both libraries on all 256 timed images (the runs above skip the eager profile passes AND the oracle): must be the same figure and the same worst image
for L in "$PREV" "X=new"; do
  env $L timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-modes --no-train-step --steps 20 --warmup 5 > gpurun_out/r6c25_ab.json 2>> gpurun_out/r6c25_ab.err
  python - "$L parity run" <<'PY' >> gpurun_out/r6c25_summary.log
import json, sys
d = json.loads(open('gpurun_out/r6c25_ab.json').read().strip().splitlines()[-1])
print(sys.argv[1][-40:], d['value'], d.get('parity'), (d.get('roofline') or {}).get('kernel'), (d.get('roofline') or {}).get('avg_launch_us'))
PY
done
tail -5 gpurun_out/r6c25_ab.err >> $S
cat $S | cut -c1-300
