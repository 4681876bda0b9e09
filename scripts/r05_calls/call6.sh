#!/bin/bash
# r05 call 6: explicit channels_last outputs of the unpadded transformer levels (FasterViT-4 level 3: coalesced window_reverse, own pool + head kernels):
# parity tests of every deploy / precise configuration, then the driver-form bench line again (the committed r05_bench_final record)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r5c6_summary.log
: > $S
timeout 1200 python -m pytest tests/test_gpu_px.py tests/test_gpu_parity.py tests/test_gpu_runtime.py tests/test_gpu_determinism.py -q -m gpu > gpurun_out/r5c6_tests.log 2>&1
echo "tests rc=$?" >> $S; tail -4 gpurun_out/r5c6_tests.log | cut -c1-300 >> $S
( time timeout 900 python bench.py --record gpurun_out/r5c6_bench_detail.json ) > gpurun_out/r5c6_bench.json 2> gpurun_out/r5c6_bench.err
echo "bench rc=$? line bytes $(tail -1 gpurun_out/r5c6_bench.json | wc -c)" >> $S
tail -4 gpurun_out/r5c6_bench.err >> $S
tail -1 gpurun_out/r5c6_bench.json >> $S
cat $S | cut -c1-5000
