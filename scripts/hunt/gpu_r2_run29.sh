#!/bin/bash
cd $GRAFT_REPO_ROOT
T=${1:-r4c}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_determinism.py -q -x -k "mlp_fused or knobs" > gpurun_out/${T}_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/${T}_pytest.log | cut -c1-300
timeout 300 python scripts/bench_stage.py ";win_mlp256=1;win_mlp256=2" > gpurun_out/${T}_bench_stage.log 2>&1; grep -v "amdgpu.ids\|UserWarning\|stage_forward(" gpurun_out/${T}_bench_stage.log | grep "level 2" | cut -c1-600
bash scripts/gpu_sweep.sh ${T} "" "-" "FVIT_TUNE_win_mlp256=2" "-" "FVIT_TUNE_win_mlp256=2"
