#!/bin/bash
# usage (on the GPU box through gpurun): bash scripts/gpu_round.sh <tag>
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
T=${1:-rX}
mkdir -p gpurun_out
S=gpurun_out/${T}_summary.log
timeout 1500 python -m pytest tests -q -m gpu -s > gpurun_out/${T}_test_gpu.log 2>&1
echo "pytest-gpu rc=$?" > $S
tail -3 gpurun_out/${T}_test_gpu.log >> $S
grep -h "err " gpurun_out/${T}_test_gpu.log >> $S
timeout 600 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc=$?" >> $S
cat gpurun_out/${T}_bench.json >> $S
timeout 300 python bench.py --mode module --no-cpu-baseline --no-secondary --steps 20 > gpurun_out/${T}_bench_module.json 2>> gpurun_out/${T}_bench.err
cat gpurun_out/${T}_bench_module.json >> $S
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_prof -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $R/gpurun_out/${T}_prof_stdout.log 2>&1
echo "rocprof rc=$?" >> $R/$S
cat $R/$S
