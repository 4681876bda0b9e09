#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -s > gpurun_out/test_gpu_all.log 2>&1
echo "pytest-gpu rc=$?" > gpurun_out/r2_summary.log
tail -3 gpurun_out/test_gpu_all.log >> gpurun_out/r2_summary.log
grep -h "err " gpurun_out/test_gpu_all.log >> gpurun_out/r2_summary.log
timeout 600 python bench.py > gpurun_out/bench_r2.json 2> gpurun_out/bench_r2.err
echo "bench rc=$?" >> gpurun_out/r2_summary.log
cat gpurun_out/bench_r2.json >> gpurun_out/r2_summary.log
timeout 300 python bench.py --no-graph --no-cpu-baseline --steps 20 > gpurun_out/bench_r2_eager.json 2>> gpurun_out/bench_r2.err
cat gpurun_out/bench_r2_eager.json >> gpurun_out/r2_summary.log
for v in fp32_cl amp_cl half_cl amp_cl_nohat half_cl_nohat; do
  timeout 300 python scripts/diag_variants.py $v 256 2>&1 | grep "img/s\|fault" >> gpurun_out/r2_summary.log
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2 -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_r2_stdout.log 2>&1
echo "rocprof rc=$?" >> $R/gpurun_out/r2_summary.log
find $R/gpurun_out/prof_r2 -name "*stats*" | head >> $R/gpurun_out/r2_summary.log
cat $R/gpurun_out/r2_summary.log
