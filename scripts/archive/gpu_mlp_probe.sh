#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "mlp or gemm_bias or glue" > gpurun_out/mlp_probe.log 2>&1
tail -2 gpurun_out/mlp_probe.log
python scripts/bench_mlp.py 54272 v0,v0a1,v1,v4,unfused >> gpurun_out/mlp_probe.log 2>&1
grep "round 2" gpurun_out/mlp_probe.log
