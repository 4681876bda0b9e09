#!/bin/bash
cd $GRAFT_REPO_ROOT
T=${1:-r4b}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/${T}_pytest_gpu.log | tail -2
timeout 400 python scripts/race_hunt4.py 12 > gpurun_out/${T}_race_hunt4.log 2>&1
echo "rc=$?"; grep -v "amdgpu.ids\|UserWarning\|stage_forward(" gpurun_out/${T}_race_hunt4.log | tail -6
bash scripts/gpu_sweep.sh ${T} "" "-" "FVIT_TUNE_win_mlp=0" "-"
bash scripts/gpu_sweep.sh ${T}s "--streams 4" "-"
bash scripts/gpu_sweep.sh ${T}t "--streams 2" "-"
