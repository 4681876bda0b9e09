#!/bin/bash
cd $GRAFT_REPO_ROOT
T=${1:-r2n}
mkdir -p gpurun_out
timeout 500 python scripts/race_hunt4.py 12 > gpurun_out/${T}_race_hunt8.log 2>&1
echo "rc=$?"; grep -v "amdgpu.ids\|UserWarning\|stage_forward(" gpurun_out/${T}_race_hunt8.log | tail -14
