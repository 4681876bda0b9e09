#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "attn_block" > gpurun_out/ab_probe.log 2>&1
tail -2 gpurun_out/ab_probe.log
python scripts/bench_attnblk.py 53 1024 >> gpurun_out/ab_probe.log 2>&1
grep "round 1" gpurun_out/ab_probe.log
