#!/bin/bash
cd $GRAFT_REPO_ROOT
T=${1:-r2q}
mkdir -p gpurun_out
timeout 500 python scripts/race_hunt13.py 6 3 > gpurun_out/${T}_race_hunt13.log 2>&1
echo "rc=$?"; grep -v "amdgpu.ids\|UserWarning\|stage_forward(" gpurun_out/${T}_race_hunt13.log | tail -60
