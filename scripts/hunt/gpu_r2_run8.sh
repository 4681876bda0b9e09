#!/bin/bash
cd $GRAFT_REPO_ROOT
T=${1:-r2i}
mkdir -p gpurun_out
timeout 400 python scripts/race_hunt5.py > gpurun_out/${T}_race_hunt5.log 2>&1
echo "rc=$?"; grep -v "amdgpu.ids" gpurun_out/${T}_race_hunt5.log | tail -30
