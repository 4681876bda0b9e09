#!/bin/bash
# ping-pong 256 x 256 GEMM, deeper request order: correctness, micro-benchmark, FasterViT-4 / any-res end to end with the knob on / off
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_precision_modes.py -q -m gpu -k "gemm_256x256" -x 2>&1 | tail -3
timeout 300 python scripts/bench_gemm.py fv4 2>&1 | grep -v amdgpu.ids | tail -18
for k in 0 1 0 1; do
FVIT_TUNE_gemm_pp=$k timeout 300 python bench.py --model faster_vit_4_224 --batch 128 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-modes --prof-steps 0 2>> gpurun_out/r3c26.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fv4 gemm_pp=$k', d['ms_per_step'], 'ms/step', d['value'], 'img/s')"
done
for k in 0 1; do
FVIT_TUNE_gemm_pp=$k timeout 300 python bench.py --model faster_vit_4_any_res --model-kwargs "{'resolution':[576,960],'window_size':[7,7,12,6],'ct_size':2}" --input-size 576x960 --batch 8 --streams 2 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-modes --prof-steps 0 2>> gpurun_out/r3c26.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('anyres gemm_pp=$k', d['ms_per_step'], 'ms/step', d['value'], 'img/s')"
done
grep -v amdgpu.ids gpurun_out/r3c26.err | tail -5
