#!/bin/bash
# r06 call 32: call 29 again with the kernel it meant to measure (scripts/bench_conv.py passed conv128_narrow = -1 for "auto", which the library reads as the 128 x 64-tile form:
# calls 29 / 30 decomposed THAT kernel): conv3x3_kernel<2,2,4> (128 x 128 tiles) alone under the ablation masks, classic and patch form
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r6c32_summary.log
: > $S
echo "== classic form (conv_patch=0), 128 x 128 tiles" >> $S
FVIT_TUNE_conv_patch=0 CONV_C=256 timeout 600 python scripts/bench_conv.py 128 56 56 gemm,gemma1,gemma2,gemma3,gemma4,gemma5,gemma7,gemma8,gemma15,gemmn,gemm >> $S 2>&1
echo "== patch form (conv_patch=1, max waste 100 %)" >> $S
FVIT_DIAG=0 FVIT_TUNE_conv_patch=1 FVIT_TUNE_conv_patch_max_waste_pct=100 CONV_C=256 timeout 600 python scripts/bench_conv.py 128 56 56 gemm,gemm >> $S 2>&1
echo "== any-res level 0 shape 8 x 144 x 240 x 256: classic, patch" >> $S
FVIT_DIAG=0 FVIT_TUNE_conv_patch=0 CONV_C=256 timeout 600 python scripts/bench_conv.py 8 144 240 gemm >> $S 2>&1
FVIT_DIAG=0 FVIT_TUNE_conv_patch=1 CONV_C=256 timeout 600 python scripts/bench_conv.py 8 144 240 gemm >> $S 2>&1
cat $S | cut -c1-200
