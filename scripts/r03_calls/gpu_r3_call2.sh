#!/bin/bash
# r03 call 2: precision-mode tests (fixed), split-hidden stage-3 MLP A/B in the bench configuration
cd $GRAFT_REPO_ROOT
T=r3c2
mkdir -p gpurun_out
S=gpurun_out/${T}_summary.log
timeout 900 python -m pytest tests/test_gpu_precision_modes.py -q -m gpu -s > gpurun_out/${T}_test_new.log 2>&1
echo "pytest-new rc=$?" > $S
grep -E "passed|failed|FAILED|err |Error" gpurun_out/${T}_test_new.log | tail -40 >> $S
summ() {
python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("  value", d["value"], "ms", d["ms_per_step"], "step_ms", d.get("step_ms", {}).get("median"), "serialized", d.get("kernel_ms_per_step_serialized"))
    r = d["roofline"]; print("  roofline", r["kernel"], r["frac"], r["avg_launch_us"], "family ms", r.get("kernel_ms_per_step_all_shapes"))
    for k in d:
        if k.startswith("parity"):
            print("  ", k, {kk: d[k].get(kk) for kk in ("logits_max_abs_err", "meets_1e-3", "images_per_s", "error")})
    for r in d["roofline_shapes"][:10]:
        print(f"  {r['kernel']:34s} wg={r['workgroups']:5d} n={r['launches_per_step']:3d} us={r['avg_launch_us']:7.2f} ms={r['ms_per_step']:.4f} frac={r['frac']}")
except Exception as e:
    print("  parse failed", e)
PY
}
for k in 1 4 2 4 1; do
  extra="--no-modes"
  if [ "$k" = "1" ] && [ ! -f gpurun_out/${T}_bench_split1.json ]; then extra=""; fi
  FVIT_TUNE_win_mlp_split=$k timeout 400 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-secondary $extra > gpurun_out/${T}_bench_split$k.json 2>> gpurun_out/${T}_bench.err
  echo "win_mlp_split=$k rc=$?" >> $S
  summ gpurun_out/${T}_bench_split$k.json >> $S
done
tail -5 gpurun_out/${T}_bench.err >> $S
cat $S
