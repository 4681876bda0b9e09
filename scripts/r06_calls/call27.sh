#!/bin/bash
# r06 call 27: two-term (x3) instances of the LONG-window attention kernel (fvit_window_attention_long_terms): kernel test vs float64, the x3 block goldens of the
# long-window tiny configurations, faster_vit_4_21k_384 in module mode + f16x3 and in the precise deploy plan (ABSOLUTE), then the rest of the x3 / px / kernel suites
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r6c27_summary.log
: > $S
timeout 1200 python -m pytest tests/test_gpu_x3.py -q -m gpu -s -x 2>&1 | grep -v "^$" | tail -60 | cut -c1-220 >> $S
timeout 900 python -m pytest tests/test_gpu_px.py -q -m gpu -s -k "precise_deploy_plan_meets" 2>&1 | tail -8 | cut -c1-220 >> $S
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "attention" 2>&1 | tail -3 >> $S
cat $S
