#!/bin/bash
# r06 call 30: s_setprio(1) around the 32 MFMAs of a K step of conv3x3_kernel (the CU's two resident workgroups alternate; the MFMA section of one should not queue
# behind the other's address arithmetic / LDS-DMA issue): variant library fastervit_amd/csrc/ab/libfvit_hip_prio.so vs the shipped one, the conv alone and FasterViT-4 end to end
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r6c30_summary.log
: > $S
for L in base prio base prio; do
  echo "== $L (conv alone, 128 x 56 x 56 x 256 -> 256)" >> $S
  FVIT_DIAG=0 FVIT_LIB_PATH=$GRAFT_REPO_ROOT/fastervit_amd/csrc/ab/libfvit_hip_$L.so CONV_C=256 timeout 300 python scripts/bench_conv.py 128 56 56 gemm,gemm 2>&1 | grep "gemm" >> $S
done
ab() {
  T="$1"; shift
  env FVIT_LIB_PATH=$GRAFT_REPO_ROOT/fastervit_amd/csrc/ab/libfvit_hip_$T.so timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-modes --no-train-step --prof-steps 0 "$@" > gpurun_out/r6c30_ab.json 2>> gpurun_out/r6c30_ab.err
  python - "$T $*" <<'PY' >> gpurun_out/r6c30_summary.log
import json, sys
try:
    d = json.loads(open('gpurun_out/r6c30_ab.json').read().strip().splitlines()[-1])
    print(f"{sys.argv[1][:110]:110s} {d['ms_per_step']:.3f} ms/step {d['value']:.1f} img/s")
except Exception as e:
    print(sys.argv[1][:110], "FAILED", e)
PY
}
F4="--model faster_vit_4_224 --batch 128 --steps 12 --warmup 3 --streams 1 --join-from 0 --inflight 2"
for rep in 1 2; do for t in base prio; do ab $t $F4; done; done
for t in base prio; do ab $t --steps 50 --warmup 10; done
tail -3 gpurun_out/r6c30_ab.err >> $S
cat $S | cut -c1-200
