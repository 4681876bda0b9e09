#!/bin/bash
# usage: bash scripts/gpu_trace.sh <tag> [bench args]  -- rocprofv3 kernel trace of a short bench run + timeline analysis
R=$GRAFT_REPO_ROOT
T=${1:-tr}; shift
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${T}_trace -o t -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --prof-steps 0 "$@" > $R/gpurun_out/${T}_trace_stdout.log 2>&1
echo "trace rc=$?"
F=$(find $R/gpurun_out/${T}_trace -name "*kernel_trace.csv" | head -1)
python $R/scripts/analyze_trace.py $F 36 > $R/gpurun_out/${T}_timeline.txt 2>&1
cat $R/gpurun_out/${T}_timeline.txt
tail -2 $R/gpurun_out/${T}_trace_stdout.log
# keep the merged output small: drop the raw trace
rm -rf $R/gpurun_out/${T}_trace
