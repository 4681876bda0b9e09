#!/bin/bash
# r06 call 29: what a K step of conv3x3_kernel<2,2,4> is made of, on FasterViT-4's level-0 shape (128 x 56 x 56 x 256 -> 256, classic K: 36 steps): the diagnosis build's
# ablation masks (1 no activation gather, 2 no weight staging, 4 no MFMA, 8 no epilogue) alone and combined -- scripts/bench_conv.py, one kernel at a time (nothing co-resident)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r6c29_summary.log
: > $S
CONV_C=256 timeout 600 python scripts/bench_conv.py 128 56 56 gemm,gemma1,gemma2,gemma3,gemma4,gemma5,gemma6,gemma7,gemma8,gemma12,gemma15,gemm >> $S 2>&1
CONV_C=128 timeout 600 python scripts/bench_conv.py 256 28 28 gemm,gemma3,gemma4,gemma7,gemma8 >> $S 2>&1
cat $S | cut -c1-220
