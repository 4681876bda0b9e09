#!/bin/bash
cd $GRAFT_REPO_ROOT
T=${1:-r2j}
mkdir -p gpurun_out
timeout 400 python scripts/race_hunt4.py 8 > gpurun_out/${T}_race_hunt4.log 2>&1
echo "rc=$?"; grep -v "amdgpu.ids\|UserWarning\|stage_forward(" gpurun_out/${T}_race_hunt4.log | tail -14
timeout 400 python scripts/race_hunt.py 16 > gpurun_out/${T}_race_hunt.log 2>&1
grep "of 15\|differs" gpurun_out/${T}_race_hunt.log | grep -v " 0 of" | head; grep -c " 0 of" gpurun_out/${T}_race_hunt.log
timeout 600 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_runtime.py tests/test_head_train.py -q -m gpu 2>&1 | tail -5
bash scripts/gpu_sweep.sh ${T} "--steps 40" - "FVIT_TUNE_mlp_stagger=2" - > /dev/null 2>&1
cat gpurun_out/${T}_sweep.log
