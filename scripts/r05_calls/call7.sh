#!/bin/bash
# r05 call 7: 128 x 128 conv tiles with a ragged last N tile for Cout % 128 == 64 (fvit_tune conv_n128_ragged): kernel tests (bitwise equal to the 128 x 64 walk)
# and A/B on both FasterViT-4 plans and any-res
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r5c7_summary.log
: > $S
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_px.py -q -m gpu -k "conv or precise" > gpurun_out/r5c7_tests.log 2>&1
echo "tests rc=$?" >> $S; tail -4 gpurun_out/r5c7_tests.log | cut -c1-300 >> $S
ab() {
  E=$1; shift
  env $E timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-modes --no-train-step --prof-steps 1 "$@" > gpurun_out/r5c7_ab.json 2>> gpurun_out/r5c7_ab.err
  python - "$E $*" <<'PY' >> gpurun_out/r5c7_summary.log
import json, sys
try:
    d = json.loads(open('gpurun_out/r5c7_ab.json').read().strip().splitlines()[-1])
    det = json.load(open('gpurun_out/bench_detail.json'))
    cv = [(x['kernel'], x['workgroups'], x['avg_launch_us'], x['launches_per_step']) for x in det.get('roofline_shapes', []) if x['kernel'].startswith('conv3x3_kernel')][:4]
    print(f"{sys.argv[1][:100]:100s} {d['ms_per_step']:.3f} ms/step {d['value']:.1f} img/s err {d['parity']['logits_max_abs_err'] if d.get('parity') else None} convs {cv}")
except Exception as e:
    print(sys.argv[1][:100], "FAILED", e)
PY
}
F4="--model faster_vit_4_224 --batch 128 --join-from 0"
AR="--model faster_vit_4_any_res --batch 8 --input-size 576x960 --join-from 0"
KW="{'resolution':[576,960],'window_size':[7,7,12,6],'ct_size':2}"
for i in 1 2; do
ab FVIT_TUNE_conv_n128_ragged=0 $F4 --operand f16x3 --precise --streams 2
ab FVIT_TUNE_conv_n128_ragged=1 $F4 --operand f16x3 --precise --streams 2
ab FVIT_TUNE_conv_n128_ragged=0 $F4 --streams 3
ab FVIT_TUNE_conv_n128_ragged=1 $F4 --streams 3
done
ab FVIT_TUNE_conv_n128_ragged=0 $AR --model-kwargs "$KW" --operand f16x3 --precise --streams 2
ab FVIT_TUNE_conv_n128_ragged=1 $AR --model-kwargs "$KW" --operand f16x3 --precise --streams 2
ab FVIT_TUNE_conv_n128_ragged=0 $AR --model-kwargs "$KW" --streams 2
ab FVIT_TUNE_conv_n128_ragged=1 $AR --model-kwargs "$KW" --streams 2
tail -3 gpurun_out/r5c7_ab.err >> $S
cat $S | cut -c1-420
