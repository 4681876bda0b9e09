#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python scripts/bench_gemm.py fv4 > gpurun_out/r3c9_gemm_fv4.log 2>&1
grep -v amdgpu gpurun_out/r3c9_gemm_fv4.log
