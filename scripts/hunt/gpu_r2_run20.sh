#!/bin/bash
# carrier-branch kernel: numerics (parity + knob-equivalence tests with the knob on), stage microbench, end to end
cd $GRAFT_REPO_ROOT
T=${1:-r3k}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_determinism.py -q -x -k "knobs" > gpurun_out/${T}_pytest_knobs.log 2>&1
echo "pytest knobs rc=$?"; tail -4 gpurun_out/${T}_pytest_knobs.log
FVIT_TUNE_ct_fused=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_runtime.py -q -x -k "faster_vit_0 or fvit0 or bench_configuration or tiny_hier" > gpurun_out/${T}_pytest_ct.log 2>&1
echo "pytest ct rc=$?"; tail -4 gpurun_out/${T}_pytest_ct.log
timeout 300 python scripts/bench_stage.py ";ct_fused=1,ct_touch=1;ct_fused=1,ct_touch=0;ct_fused=1,ct_touch=1,ct_variant=1;ct_fused=1,ct_touch=1,ct_variant=2" > gpurun_out/${T}_bench_stage.log 2>&1; grep -v "amdgpu.ids\|UserWarning" gpurun_out/${T}_bench_stage.log | tail -9
bash scripts/gpu_sweep.sh ${T} "" "-" "FVIT_TUNE_ct_fused=1" "FVIT_TUNE_ct_fused=1 FVIT_TUNE_ct_touch=0" "FVIT_TUNE_ct_fused=1 FVIT_TUNE_ct_variant=1" "-" "FVIT_TUNE_ct_fused=1"
