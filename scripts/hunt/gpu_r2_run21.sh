#!/bin/bash
cd $GRAFT_REPO_ROOT
T=${1:-r3o}
mkdir -p gpurun_out
timeout 300 python scripts/bench_stage.py "${2:-;ct_fused=1,ct_touch=0;ct_fused=1,ct_touch=1}" > gpurun_out/${T}_bench_stage.log 2>&1; grep -v "amdgpu.ids\|UserWarning\|stage_forward(" gpurun_out/${T}_bench_stage.log | cut -c1-900 | tail -12
