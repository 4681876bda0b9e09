#!/bin/bash
# usage: bash scripts/gpu_quick.sh <tag> [pytest -k expr] [extra env for a second bench, e.g. FVIT_TUNE_conv64_variant=1]
cd $GRAFT_REPO_ROOT
T=${1:-q}
mkdir -p gpurun_out
S=gpurun_out/${T}_summary.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "${2:-mlp or glue or gemm_bias}" > gpurun_out/${T}_test_k.log 2>&1
echo "kernels rc=$?" > $S; tail -15 gpurun_out/${T}_test_k.log >> $S
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s > gpurun_out/${T}_test_p.log 2>&1
echo "parity rc=$?" >> $S; tail -4 gpurun_out/${T}_test_p.log >> $S; grep -h "err " gpurun_out/${T}_test_p.log >> $S
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc=$?" >> $S; cat gpurun_out/${T}_bench.json >> $S; tail -3 gpurun_out/${T}_bench.err >> $S
if [ -n "$3" ]; then
  env $3 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/${T}_bench_alt.json 2>> gpurun_out/${T}_bench.err
  echo "alt bench ($3) rc=$?" >> $S; cat gpurun_out/${T}_bench_alt.json >> $S
fi
cat $S
