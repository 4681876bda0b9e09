#!/bin/bash
cd $GRAFT_REPO_ROOT
T=${1:-r4l}
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_determinism.py -q -x -k "attn_block_fused or knobs18 or knobs19 or knobs20" > gpurun_out/${T}_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/${T}_pytest.log | cut -c1-300
timeout 200 python scripts/bench_stage.py ";win_fused256=1" > gpurun_out/${T}_bench_stage.log 2>&1; grep -v "amdgpu.ids\|UserWarning\|stage_forward(" gpurun_out/${T}_bench_stage.log | grep "level 2" | cut -c1-700
bash scripts/gpu_sweep.sh ${T} "" "-" "FVIT_TUNE_win_fused256=1" "-" "FVIT_TUNE_win_fused256=1"
