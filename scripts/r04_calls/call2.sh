#!/bin/bash
# r04 call 2: x3 operand modes (two-term activations): kernel / block / model tests, regression of the touched kernels, bench with the timed precise legs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_x3.py -q -m gpu -s 2>&1 | grep -v "^$" | tail -60 > gpurun_out/r4c2_x3.log; tail -45 gpurun_out/r4c2_x3.log
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_precision_modes.py -q -m gpu -x -k "gemm or layernorm or attention" 2>&1 | tail -4
( time timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > gpurun_out/r4c2_bench.json 2> gpurun_out/r4c2_bench.err
tail -1 gpurun_out/r4c2_bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['parity'])
for s in d.get('secondary',[]): print(json.dumps(s))
"
tail -5 gpurun_out/r4c2_bench.err
