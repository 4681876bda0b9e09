#!/bin/bash
# r06 call 26: the whole library under other LLVM machine-scheduler settings (never tried in r01-r05): -mllvm -amdgpu-sched-strategy=max-ilp | max-memory-clause,
# -mllvm -enable-post-misched=0; variants built to fastervit_amd/csrc/ab/libfvit_hip_<tag>.so (uncommitted build products), selected with FVIT_LIB_PATH, interleaved in one box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r6c26_summary.log
: > $S
ab() {
  T="$1"; shift
  E="X=base"; [ "$T" != "base" ] && E="FVIT_LIB_PATH=$GRAFT_REPO_ROOT/fastervit_amd/csrc/ab/libfvit_hip_$T.so"
  env $E timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-modes --no-train-step --prof-steps 0 "$@" > gpurun_out/r6c26_ab.json 2>> gpurun_out/r6c26_ab.err
  python - "$T $*" <<'PY' >> gpurun_out/r6c26_summary.log
import json, sys
try:
    d = json.loads(open('gpurun_out/r6c26_ab.json').read().strip().splitlines()[-1])
    print(f"{sys.argv[1][:110]:110s} {d['ms_per_step']:.3f} ms/step {d['value']:.1f} img/s")
except Exception as e:
    print(sys.argv[1][:110], "FAILED", e)
PY
}
F4="--model faster_vit_4_224 --batch 128 --steps 12 --warmup 3 --streams 1 --join-from 0 --inflight 2"
for rep in 1 2; do
  for t in base ilp memcl nopost; do ab $t --steps 50 --warmup 10; done
done
for t in base ilp memcl nopost; do ab $t $F4; done
tail -5 gpurun_out/r6c26_ab.err >> $S
cat $S | cut -c1-300
