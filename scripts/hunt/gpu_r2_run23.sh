#!/bin/bash
cd $GRAFT_REPO_ROOT
T=${1:-r3s}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -s > gpurun_out/${T}_pytest_parity.log 2>&1
echo "pytest rc=$?"; grep -v "amdgpu.ids" gpurun_out/${T}_pytest_parity.log | tail -8
