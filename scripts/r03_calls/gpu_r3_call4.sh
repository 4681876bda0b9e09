#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python scripts/timeline_winmlp.py > gpurun_out/r3c4_timeline.log 2>&1
echo "timeline rc=$?"
grep -v amdgpu gpurun_out/r3c4_timeline.log
