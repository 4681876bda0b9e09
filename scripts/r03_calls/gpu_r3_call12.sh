#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "stem" 2>&1 | tail -3
timeout 300 python scripts/bench_stem.py 86 > gpurun_out/r3c12_stem.log 2>&1
grep -v amdgpu gpurun_out/r3c12_stem.log
