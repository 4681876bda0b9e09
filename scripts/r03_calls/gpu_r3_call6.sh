#!/bin/bash
# marginal cost of each kernel family inside the 3 concurrent stream shards: bench step time with the family's launches skipped
cd $GRAFT_REPO_ROOT
T=r3c6
mkdir -p gpurun_out
S=gpurun_out/${T}_summary.log
echo "# ablate_skip bits: 1 winmlp<256>, 2 winmlp<512>, 4 winblk, 8 attnblk, 16 ctblk, 32 conv3x3 implicit GEMM, 64 halo conv, 128 fused stem; 255 = all" > $S
for k in 0 1 2 4 8 16 32 64 128 0 255 31; do
  FVIT_TUNE_ablate_skip=$k timeout 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary --no-modes --prof-steps 0 > gpurun_out/${T}_b$k.json 2>> gpurun_out/${T}.err
  echo "ablate_skip=$k: $(python -c "import json;d=json.load(open('gpurun_out/${T}_b$k.json'));print(d['ms_per_step'], 'ms/step', d['value'], 'img/s')")" >> $S
done
timeout 200 python scripts/timeline_winmlp.py > gpurun_out/${T}_timeline.log 2>&1
grep -v amdgpu gpurun_out/${T}_timeline.log | head -70 >> $S
cat $S
