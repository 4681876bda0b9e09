#!/bin/bash
cd $GRAFT_REPO_ROOT
T=${1:-r3r}
mkdir -p gpurun_out
for B in 86 22 256; do timeout 200 python scripts/bench_ctblk.py $B >> gpurun_out/${T}_bench_ctblk.log 2>&1; done
grep -v "amdgpu.ids\|UserWarning" gpurun_out/${T}_bench_ctblk.log | tail -26
