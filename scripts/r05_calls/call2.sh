#!/bin/bash
# r05 call 2: full GPU test suite on the current tree (diag split, px kernels, precise plan, dual x3 GEMM) + smoke + the driver-form bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r5c2_summary.log
: > $S
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r5c2_test_gpu.log 2>&1
echo "pytest-gpu rc=$?" >> $S
tail -15 gpurun_out/r5c2_test_gpu.log >> $S
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> $S 2>&1
( time timeout 900 python bench.py ) > gpurun_out/r5c2_bench.json 2> gpurun_out/r5c2_bench.err
echo "bench rc=$? line bytes $(tail -1 gpurun_out/r5c2_bench.json | wc -c)" >> $S
tail -4 gpurun_out/r5c2_bench.err >> $S
cp gpurun_out/bench_detail.json gpurun_out/r5c2_bench_detail.json
tail -1 gpurun_out/r5c2_bench.json >> $S
cat $S | cut -c1-6000
