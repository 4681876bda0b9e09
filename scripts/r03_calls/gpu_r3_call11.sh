#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python scripts/bench_stem.py 86 > gpurun_out/r3c11_stem.log 2>&1
grep -v amdgpu gpurun_out/r3c11_stem.log
