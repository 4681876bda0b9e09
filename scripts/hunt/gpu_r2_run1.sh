#!/bin/bash
# round 2, GPU call 1: probes + stagger A/B + full GPU test suite (host-runtime refactor)
cd $GRAFT_REPO_ROOT
T=${1:-r2a}
mkdir -p gpurun_out
timeout 120 scripts/probes/dma_probe2 > gpurun_out/${T}_dma_probe2.log 2>&1
echo "dma_probe2 rc=$?"
timeout 240 python scripts/bench_gemm.py > gpurun_out/${T}_bench_gemm.log 2>&1
echo "bench_gemm rc=$?"
for M in 1360 4096 18020; do
  timeout 100 python scripts/bench_mlp.py $M v0s0,v0,v0s2,unfused >> gpurun_out/${T}_bench_mlp.log 2>&1
done
echo "bench_mlp rc=$?"
for NW in 85 256; do
  timeout 100 python scripts/bench_attnblk.py 16 $NW a0,a0g,unfused >> gpurun_out/${T}_bench_attnblk.log 2>&1
done
timeout 100 python scripts/bench_attnblk.py 53 340 a0,a0g,unfused >> gpurun_out/${T}_bench_attnblk.log 2>&1
echo "bench_attnblk rc=$?"
bash scripts/gpu_sweep.sh ${T} "--steps 40" - "FVIT_TUNE_gemm_stagger=1" "FVIT_TUNE_gemm_stagger=1 FVIT_TUNE_ab_stagger=1 FVIT_TUNE_mlp_stagger=2" > /dev/null 2>&1
bash scripts/gpu_sweep.sh ${T}s1 "--steps 40 --streams 1" - "FVIT_TUNE_gemm_stagger=1 FVIT_TUNE_ab_stagger=1 FVIT_TUNE_mlp_stagger=2" > /dev/null 2>&1
timeout 900 python -m pytest tests -q -m gpu -s -x > gpurun_out/${T}_test_gpu.log 2>&1
echo "pytest-gpu rc=$?"
tail -n 40 gpurun_out/${T}_test_gpu.log | cut -c1-400
grep -h "err " gpurun_out/${T}_test_gpu.log | tail -40
cat gpurun_out/${T}_sweep.log gpurun_out/${T}s1_sweep.log
