#!/bin/bash
# r05 call 3: the tests behind call 2's stop (-x) and those touched since (GEMM / conv-px epilogues dispatched once, avgpool + head tail, eval-mode
# routing), then FasterViT-4 / any-res in both configurations and the training-step entry
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r5c3_summary.log
: > $S
timeout 1200 python -m pytest tests/test_gpu_runtime.py tests/test_gpu_x3.py tests/test_gpu_px.py tests/test_head_train.py tests/test_sharded_validate.py tests/test_gpu_precision_modes.py -q -m gpu > gpurun_out/r5c3_tests_a.log 2>&1
echo "tests A rc=$?" >> $S; tail -6 gpurun_out/r5c3_tests_a.log | cut -c1-300 >> $S
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -k "gemm or conv or stem or fvit4 or fvit0 or layernorm2d" > gpurun_out/r5c3_tests_b.log 2>&1
echo "tests B rc=$?" >> $S; tail -4 gpurun_out/r5c3_tests_b.log | cut -c1-300 >> $S
ab() {
  E=$1; shift
  env $E timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-modes --prof-steps 1 "$@" > gpurun_out/r5c3_ab.json 2>> gpurun_out/r5c3_ab.err
  python - "$E $*" <<'PY' >> gpurun_out/r5c3_summary.log
import json, sys
try:
    d = json.loads(open('gpurun_out/r5c3_ab.json').read().strip().splitlines()[-1])
    r = d.get('roofline') or {}
    print(f"{sys.argv[1][:140]:140s} {d['ms_per_step']:.3f} ms/step {d['value']:.1f} img/s err {d['parity']['logits_max_abs_err'] if d.get('parity') else None} dom {r.get('kernel')} {r.get('avg_launch_us')} us")
except Exception as e:
    print(sys.argv[1][:140], "FAILED", e)
PY
}
F4="--model faster_vit_4_224 --batch 128 --join-from 0"
AR="--model faster_vit_4_any_res --batch 8 --input-size 576x960 --join-from 0"
KW="{'resolution':[576,960],'window_size':[7,7,12,6],'ct_size':2}"
ab X=1 $F4 --operand f16x3 --precise --streams 2
ab X=1 $F4 --operand f16 --streams 3
ab X=1 $AR --model-kwargs "$KW" --operand f16x3 --precise --streams 2
ab X=1 $AR --model-kwargs "$KW" --operand f16 --streams 2
ab X=1
python - <<'PY' >> gpurun_out/r5c3_summary.log 2>&1
import json, sys, argparse, torch
sys.path.insert(0, '.')
import bench
r = bench.run_train_step(argparse.Namespace(), torch.device('cuda', 0))
print("train_step:", json.dumps(r)[:600])
PY
tail -3 gpurun_out/r5c3_ab.err >> $S
cat $S | cut -c1-420
