#!/bin/bash
# band conv kernel: do the workgroups collide on the same L2 lines of the weight stream?  replicated weights, unit u reads copy u % N
cd $GRAFT_REPO_ROOT
for n in 1 2 8 16; do
  echo "== weight copies $n"
  FVIT_TUNE_conv_band_copies=$n timeout 300 python scripts/bench_conv128.py 86 256 2>&1 | grep -v amdgpu.ids | grep -v "implicit" | grep -v "^   \|timeline"
done
