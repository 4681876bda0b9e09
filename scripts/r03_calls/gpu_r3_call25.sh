#!/bin/bash
# ping-pong 256 x 256 GEMM: correctness (bitwise vs the 128 tile) and micro-benchmark on the FasterViT-4 shapes
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_precision_modes.py -q -m gpu -k "gemm_256x256" -x 2>&1 | tail -6
timeout 300 python scripts/bench_gemm.py fv4 2>&1 | grep -v amdgpu.ids | tail -20
