#!/bin/bash
cd $GRAFT_REPO_ROOT
T=r3c3
mkdir -p gpurun_out
S=gpurun_out/${T}_summary.log
timeout 300 python scripts/timeline_winmlp.py > gpurun_out/${T}_timeline.log 2>&1
echo "timeline rc=$?" > $S
cat gpurun_out/${T}_timeline.log >> $S
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "attn_block_fused" > gpurun_out/${T}_test.log 2>&1
echo "pytest attn_block rc=$?" >> $S
tail -4 gpurun_out/${T}_test.log >> $S
for k in 2; do
  FVIT_TUNE_win_blk_split=$k timeout 400 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-secondary --no-modes > gpurun_out/${T}_bench_blk$k.json 2>> gpurun_out/${T}_bench.err
  python - gpurun_out/${T}_bench_blk$k.json >> $S <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("win_blk_split=2 value", d["value"], "ms", d["ms_per_step"])
for r in d["roofline_shapes"][:6]:
    print(f"  {r['kernel']:34s} wg={r['workgroups']:5d} n={r['launches_per_step']:3d} us={r['avg_launch_us']:7.2f} ms={r['ms_per_step']:.4f} frac={r['frac']}")
PY
done
cat $S
