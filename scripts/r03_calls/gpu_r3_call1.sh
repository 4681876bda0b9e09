#!/bin/bash
# r03 call 1: new precision-mode / 256-tile tests, full GPU suite, FasterViT-4 GEMM microbench, default bench line, gemm256 A/B
cd $GRAFT_REPO_ROOT
T=r3c1
mkdir -p gpurun_out
S=gpurun_out/${T}_summary.log
timeout 600 python -m pytest tests/test_gpu_precision_modes.py -q -m gpu -s -x > gpurun_out/${T}_test_new.log 2>&1
echo "pytest-new rc=$?" > $S
tail -15 gpurun_out/${T}_test_new.log >> $S
timeout 300 python scripts/bench_gemm.py fv4 > gpurun_out/${T}_gemm_fv4.log 2>&1
cat gpurun_out/${T}_gemm_fv4.log >> $S
timeout 900 python -m pytest tests -q -m gpu -s --deselect tests/test_gpu_precision_modes.py > gpurun_out/${T}_test_gpu.log 2>&1
echo "pytest-gpu rc=$?" >> $S
tail -5 gpurun_out/${T}_test_gpu.log >> $S
grep -h "err " gpurun_out/${T}_test_gpu.log gpurun_out/${T}_test_new.log | tail -60 >> $S
timeout 700 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc=$?" >> $S
python - <<'PY' >> $S
import json
try:
    d = json.load(open("gpurun_out/r3c1_bench.json"))
    print("value", d["value"], "ms", d["ms_per_step"], "step_ms", d.get("step_ms"))
    print("roofline", json.dumps(d["roofline"])[:900])
    for k in d:
        if k.startswith("parity"):
            print(k, json.dumps(d[k])[:400])
    print("cpu", json.dumps(d.get("cpu_baseline"))[:300])
    for s in d.get("secondary", []):
        print("secondary", json.dumps({k: v for k, v in s.items() if k != "roofline"})[:600])
        print("   roof", json.dumps(s.get("roofline"))[:500])
    for r in d["roofline_shapes"][:14]:
        print(f"{r['kernel']:34s} wg={r['workgroups']:5d} n={r['launches_per_step']:3d} us={r['avg_launch_us']:7.2f} ms={r['ms_per_step']:.4f} frac={r['frac']}")
except Exception as e:
    print("bench parse failed", e)
PY
for k in 0 192; do
FVIT_TUNE_gemm256_min_tiles=$k timeout 300 python bench.py --model faster_vit_4_224 --batch 128 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --prof-steps 0 > gpurun_out/${T}_fv4_gemm256_$k.json 2>> gpurun_out/${T}_bench.err
echo "fv4 gemm256_min_tiles=$k: $(python -c "import json;d=json.load(open('gpurun_out/${T}_fv4_gemm256_$k.json'));print(d['value'], d['ms_per_step'])")" >> $S
done
tail -5 gpurun_out/${T}_bench.err >> $S
cat $S
