#!/bin/bash
cd $GRAFT_REPO_ROOT
T=${1:-r3t}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_determinism.py -q -x -k "ct_block or knobs or bench_configuration" > gpurun_out/${T}_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/${T}_pytest.log
timeout 200 python scripts/bench_ctblk.py 86 > gpurun_out/${T}_bench_ctblk.log 2>&1; grep -v "amdgpu.ids\|UserWarning" gpurun_out/${T}_bench_ctblk.log | head -4
bash scripts/gpu_sweep.sh ${T} "" "-" "FVIT_TUNE_ct_fused=0" "-" "FVIT_TUNE_ct_fused=0"
