#!/bin/bash
# round 2, GPU call 3: weight-ring depth (MLP ring 4, GEMM ring 3/4), attention block with double-buffered qkv slices, chunk stagger
cd $GRAFT_REPO_ROOT
T=${1:-r2c}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_determinism.py -q -x -k "mlp or attn_block or gemm or knobs or repeatable" > gpurun_out/${T}_test_k.log 2>&1
echo "kernel tests rc=$?"; tail -3 gpurun_out/${T}_test_k.log
for M in 1360 18020; do
  FVIT_TUNE_mlp_ring4_max_grid=0 timeout 100 python scripts/bench_mlp.py $M v0,v0s0,v0a2 >> gpurun_out/${T}_bench_mlp_ring2.log 2>&1
  timeout 100 python scripts/bench_mlp.py $M v0,v0s0,v0a2,unfused >> gpurun_out/${T}_bench_mlp_ring4.log 2>&1
done
echo "--- mlp ring2"; grep "round 1" gpurun_out/${T}_bench_mlp_ring2.log; echo "--- mlp ring4"; grep "round 1" gpurun_out/${T}_bench_mlp_ring4.log
timeout 100 python scripts/bench_attnblk.py 53 340 a0,a0v1,a0v2,a1,a1v2,unfused >> gpurun_out/${T}_bench_attnblk.log 2>&1
timeout 100 python scripts/bench_attnblk.py 53 1024 a0,a0v1,a0v2 >> gpurun_out/${T}_bench_attnblk.log 2>&1
grep "round 1" gpurun_out/${T}_bench_attnblk.log
for R in 2 3 4; do
  echo "--- gemm ring $R" >> gpurun_out/${T}_bench_gemm.log
  FVIT_TUNE_gemm_ring=$R timeout 120 python scripts/bench_gemm.py ring >> gpurun_out/${T}_bench_gemm.log 2>&1
done
cat gpurun_out/${T}_bench_gemm.log | grep -v amdgpu.ids
bash scripts/gpu_sweep.sh ${T} "--steps 40" - "FVIT_TUNE_ab_variant=2" "FVIT_TUNE_mlp_ring4_max_grid=0" "FVIT_TUNE_gemm_ring=3" "FVIT_TUNE_gemm_ring=4" "FVIT_TUNE_ab_variant=2 FVIT_TUNE_gemm_ring=4" "FVIT_TUNE_mlp_stagger=0" > /dev/null 2>&1
bash scripts/gpu_sweep.sh ${T}s2 "--steps 40 --streams 2" - "FVIT_TUNE_ab_variant=2 FVIT_TUNE_gemm_ring=4" > /dev/null 2>&1
bash scripts/gpu_sweep.sh ${T}s4 "--steps 40 --streams 4" - "FVIT_TUNE_ab_variant=2 FVIT_TUNE_gemm_ring=4" > /dev/null 2>&1
cat gpurun_out/${T}_sweep.log gpurun_out/${T}s2_sweep.log gpurun_out/${T}s4_sweep.log
timeout 400 python bench.py --steps 30 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc=$?"; tail -c 3000 gpurun_out/${T}_bench.json; tail -5 gpurun_out/${T}_bench.err
