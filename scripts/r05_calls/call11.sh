#!/bin/bash
# r05 call 11: the full GPU suite + smoke + driver-form bench line on the final tree (after Dropout / attn_drop)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r5c11_summary.log
: > $S
timeout 2400 python -m pytest tests -q -m gpu -s > gpurun_out/r5c11_test_gpu.log 2>&1
echo "pytest-gpu rc=$?" >> $S
tail -3 gpurun_out/r5c11_test_gpu.log | cut -c1-300 >> $S
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> $S 2>&1
( time timeout 900 python bench.py --record gpurun_out/r5c11_bench_detail.json ) > gpurun_out/r5c11_bench.json 2> gpurun_out/r5c11_bench.err
echo "bench rc=$? line bytes $(tail -1 gpurun_out/r5c11_bench.json | wc -c)" >> $S
tail -1 gpurun_out/r5c11_bench.json >> $S
cat $S | cut -c1-4500
