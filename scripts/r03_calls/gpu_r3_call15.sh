#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_determinism.py -q -m gpu -k "partition or token_init or tiny or fvit0_224 or deploy or repeatable or anyres" 2>&1 | tail -4
for i in 1 2; do
timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary --no-modes --prof-steps 0 > gpurun_out/r3c15_tmp.json 2>> gpurun_out/r3c15.err
python -c "import json;d=json.load(open('gpurun_out/r3c15_tmp.json'));print('bench', d['ms_per_step'], 'ms/step', d['value'], 'img/s')"
done
tail -2 gpurun_out/r3c15.err
