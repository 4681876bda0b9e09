#!/bin/bash
# fused-MLP variant 6 (2 waves x 32 rows): numerics, microbench, end to end
cd $GRAFT_REPO_ROOT
T=${1:-r3i}
mkdir -p gpurun_out
FVIT_TUNE_mlp_variant=6 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -x -k "mlp or faster_vit_0" > gpurun_out/${T}_pytest_v6.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/${T}_pytest_v6.log
timeout 300 python scripts/bench_mlp.py 18232 v0s0,v6s0,v3s0,v0s0,v6s0,unfused > gpurun_out/${T}_bench_mlp.log 2>&1; grep -v "amdgpu.ids\|UserWarning" gpurun_out/${T}_bench_mlp.log | tail -8
bash scripts/gpu_sweep.sh ${T} "" "-" "FVIT_TUNE_mlp_variant=6" "-" "FVIT_TUNE_mlp_variant=6"
