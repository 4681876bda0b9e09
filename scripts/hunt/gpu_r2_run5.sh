#!/bin/bash
cd $GRAFT_REPO_ROOT
T=${1:-r2f}
mkdir -p gpurun_out
timeout 300 python scripts/race_hunt2.py > gpurun_out/${T}_race_hunt2.log 2>&1
echo "rc=$?"; grep -v "amdgpu.ids\|UserWarning\|stage_forward" gpurun_out/${T}_race_hunt2.log | tail -30
