#!/bin/bash
cd $GRAFT_REPO_ROOT
T=${1:-r4m}
mkdir -p gpurun_out
timeout 330 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/${T}_pytest_gpu.log | tail -2
