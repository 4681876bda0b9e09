#!/bin/bash
# usage: bash scripts/gpu_conv_probe.sh <tag>  -- conv/token kernel tests, conv micro-benchmark, parity, bench
cd $GRAFT_REPO_ROOT
T=${1:-cv}
mkdir -p gpurun_out
S=gpurun_out/${T}_summary.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "conv or token or stem" > gpurun_out/${T}_test_k.log 2>&1
echo "kernels rc=$?" > $S; tail -8 gpurun_out/${T}_test_k.log >> $S
timeout 300 python scripts/bench_conv.py 85 56 56 > gpurun_out/${T}_bench_conv.log 2>&1
timeout 300 python scripts/bench_conv.py 256 56 56 gemm,halo >> gpurun_out/${T}_bench_conv.log 2>&1
cat gpurun_out/${T}_bench_conv.log >> $S
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s > gpurun_out/${T}_test_p.log 2>&1
echo "parity rc=$?" >> $S; tail -4 gpurun_out/${T}_test_p.log >> $S; grep -h "err " gpurun_out/${T}_test_p.log >> $S
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc=$?" >> $S; cat gpurun_out/${T}_bench.json >> $S; tail -3 gpurun_out/${T}_bench.err >> $S
FVIT_TUNE_conv_halo=0 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/${T}_bench_nohalo.json 2>> gpurun_out/${T}_bench.err
echo "bench (conv_halo=0) rc=$?" >> $S; cat gpurun_out/${T}_bench_nohalo.json >> $S
cat $S
