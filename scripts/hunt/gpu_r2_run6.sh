#!/bin/bash
cd $GRAFT_REPO_ROOT
T=${1:-r2g}
mkdir -p gpurun_out
timeout 300 python scripts/race_hunt3.py > gpurun_out/${T}_race_hunt3.log 2>&1
echo "rc=$?"; grep -v "amdgpu.ids\|UserWarning\|stage_forward(" gpurun_out/${T}_race_hunt3.log | tail -30
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "ln_gemm" 2>&1 | tail -3
bash scripts/gpu_sweep.sh ${T} "--steps 40" - "FVIT_TUNE_ln_gemm=0" "FVIT_TUNE_pe_preadd=0" > /dev/null 2>&1
cat gpurun_out/${T}_sweep.log
