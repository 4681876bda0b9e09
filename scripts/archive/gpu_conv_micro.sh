#!/bin/bash
# usage: bash scripts/gpu_conv_micro.sh <tag> [bench_conv variants]  -- conv kernel tests + conv micro-benchmark only
cd $GRAFT_REPO_ROOT
T=${1:-cm}
mkdir -p gpurun_out
S=gpurun_out/${T}_summary.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "conv" > gpurun_out/${T}_test_k.log 2>&1
echo "kernels rc=$?" > $S; tail -5 gpurun_out/${T}_test_k.log >> $S
timeout 300 python scripts/bench_conv.py 85 56 56 ${2:-gemm,halo} > gpurun_out/${T}_bench_conv.log 2>&1
timeout 300 python scripts/bench_conv.py 256 56 56 ${2:-gemm,halo} >> gpurun_out/${T}_bench_conv.log 2>&1
grep -v amdgpu.ids gpurun_out/${T}_bench_conv.log >> $S
cat $S
