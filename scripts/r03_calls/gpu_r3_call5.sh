#!/bin/bash
cd $GRAFT_REPO_ROOT
T=r3c5
mkdir -p gpurun_out
S=gpurun_out/${T}_summary.log
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_precision_modes.py -q -m gpu -x -k "mlp_fused or attn_block_fused or gemm or conv3x3 or win_mlp or ct_block" > gpurun_out/${T}_test.log 2>&1
echo "pytest rc=$?" > $S
tail -4 gpurun_out/${T}_test.log >> $S
timeout 300 python scripts/timeline_winmlp.py > gpurun_out/${T}_timeline.log 2>&1
grep -v amdgpu gpurun_out/${T}_timeline.log | grep -A18 "M=18240\|M=4214" | grep -v "raw stamps" >> $S
timeout 400 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-modes > gpurun_out/${T}_bench.json 2>> gpurun_out/${T}_bench.err
python - gpurun_out/${T}_bench.json >> $S <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("value", d["value"], "ms", d["ms_per_step"], "serialized", d.get("kernel_ms_per_step_serialized"))
for r in d["roofline_shapes"][:16]:
    print(f"  {r['kernel']:34s} wg={r['workgroups']:5d} n={r['launches_per_step']:3d} us={r['avg_launch_us']:7.2f} ms={r['ms_per_step']:.4f} frac={r['frac']}")
for s in d.get("secondary", []):
    print("secondary", s.get("workload"), s.get("value"), s.get("ms_per_step"), s.get("parity"), s.get("error"))
    r = s.get("roofline") or {}
    print("   roof", r.get("kernel"), r.get("frac"), r.get("avg_launch_us"), r.get("kernel_ms_per_step_all_shapes"))
PY
tail -3 gpurun_out/${T}_bench.err >> $S
cat $S
