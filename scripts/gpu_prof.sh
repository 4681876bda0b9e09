#!/bin/bash
# usage: bash scripts/gpu_prof.sh <tag>  -- rocprofv3 kernel trace + stats of a short bench run
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
T=${1:-p}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_prof -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/${T}_prof_stdout.log 2>&1
echo "rocprof rc=$?"
ls $R/gpurun_out/${T}_prof
