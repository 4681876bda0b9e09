#!/bin/bash
# band conv kernel: B-fragment batch / weight-ring depth variants (libraries prebuilt under variants/), micro-benchmark + timeline each
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp fastervit_amd/csrc/libfvit_hip.so /tmp/libfvit_hip_default.so
for v in d8_p3 d8_p5 d6_p3 d6_p2; do
  cp variants/libfvit_hip_$v.so fastervit_amd/csrc/libfvit_hip.so
  echo "== variant $v"
  timeout 300 python scripts/bench_conv128.py 86 2>&1 | grep -v amdgpu.ids | grep -v "implicit" | tail -22
done
cp /tmp/libfvit_hip_default.so fastervit_amd/csrc/libfvit_hip.so
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "c128_band" 2>&1 | tail -2
