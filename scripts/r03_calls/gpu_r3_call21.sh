#!/bin/bash
# band conv kernel: phase timeline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python scripts/bench_conv128.py 86 2>&1 | grep -v amdgpu.ids
