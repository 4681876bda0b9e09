#!/bin/bash
# round 2, GPU call 2: ILP rewrite of the fused MLP / attention-block loops: kernel tests, microbenches, end-to-end A/B of the
# fused-kernel thresholds, full GPU suite
cd $GRAFT_REPO_ROOT
T=${1:-r2b}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "mlp or attn_block or gemm" > gpurun_out/${T}_test_k.log 2>&1
echo "kernel tests rc=$?"; tail -3 gpurun_out/${T}_test_k.log
for M in 1360 4096 18020; do
  timeout 100 python scripts/bench_mlp.py $M v0,unfused >> gpurun_out/${T}_bench_mlp.log 2>&1
done
for NW in 85 256; do
  timeout 100 python scripts/bench_attnblk.py 16 $NW a0,unfused >> gpurun_out/${T}_bench_attnblk.log 2>&1
done
timeout 100 python scripts/bench_attnblk.py 53 340 a0,unfused >> gpurun_out/${T}_bench_attnblk.log 2>&1
grep "round 1" gpurun_out/${T}_bench_mlp.log gpurun_out/${T}_bench_attnblk.log
bash scripts/gpu_sweep.sh ${T} "--steps 40" - "FVIT_TUNE_attn_fused_min_rows=0 FVIT_TUNE_mlp_fused_min_rows=0" "FVIT_TUNE_attn_fused_min_rows=0 FVIT_TUNE_mlp_fused_min_rows=0 FVIT_TUNE_mlp_fused512_min_rows=0 FVIT_TUNE_attn_fused512_min_rows=0" "FVIT_TUNE_mlp_fused512_min_rows=0 FVIT_TUNE_attn_fused512_min_rows=0" "FVIT_TUNE_mlp_fused_min_rows=0" > /dev/null 2>&1
bash scripts/gpu_sweep.sh ${T}s1 "--steps 40 --streams 1" - "FVIT_TUNE_attn_fused_min_rows=0 FVIT_TUNE_mlp_fused_min_rows=0" > /dev/null 2>&1
bash scripts/gpu_sweep.sh ${T}s2 "--steps 40 --streams 2" - "FVIT_TUNE_attn_fused_min_rows=0 FVIT_TUNE_mlp_fused_min_rows=0" > /dev/null 2>&1
cat gpurun_out/${T}_sweep.log gpurun_out/${T}s1_sweep.log gpurun_out/${T}s2_sweep.log
timeout 900 python -m pytest tests -q -m gpu -s > gpurun_out/${T}_test_gpu.log 2>&1
echo "pytest-gpu rc=$?"
tail -n 15 gpurun_out/${T}_test_gpu.log | cut -c1-300
grep -h "err " gpurun_out/${T}_test_gpu.log | tail -45
