#!/bin/bash
# r04 call 6: split-K of the small-grid residual GEMMs: kernel test, FasterViT-4 / any-res goldens + determinism, throughput A/B (FVIT_TUNE_gemm_splitk=0/1)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_x3.py tests/test_gpu_parity.py tests/test_gpu_precision_modes.py tests/test_gpu_determinism.py -q -m gpu -s -k "split_k or fvit4 or anyres or tiny or repeatab or determin" 2>&1 | grep -E "passed|failed|Error|assert|logits max-abs" | cut -c1-200 | tail -30
ab() {
  E=$1; shift
  env $E timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-modes --prof-steps 1 "$@" > gpurun_out/r4c6_ab.json 2>> gpurun_out/r4c6_ab.err
  python - "$E $*" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/r4c6_ab.json').read().strip().splitlines()[-1])
print(f"{sys.argv[1][:110]:110s} {d['ms_per_step']:.4f} ms/step {d['value']:.1f} img/s err {d['parity']['logits_max_abs_err'] if d.get('parity') else None}")
PY
}
F4="--model faster_vit_4_224 --batch 128 --streams 3 --join-from 0"
AR="--model faster_vit_4_any_res --batch 8 --input-size 576x960 --streams 2 --join-from 0"
KW="{'resolution':[576,960],'window_size':[7,7,12,6],'ct_size':2}"
for i in 1 2; do
ab FVIT_TUNE_gemm_splitk=1 $F4
ab FVIT_TUNE_gemm_splitk=0 $F4
done
ab FVIT_TUNE_gemm_splitk_slots=920 $F4
ab FVIT_TUNE_gemm_splitk_slots=300 $F4
ab FVIT_TUNE_gemm_splitk=1 $AR --model-kwargs "$KW"
ab FVIT_TUNE_gemm_splitk=0 $AR --model-kwargs "$KW"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_detail.json'))
for r in d['roofline_shapes'][:14]: print(f"{r['kernel']:34s} wg={r['workgroups']:5d} n={r['launches_per_step']:3d} us={r['avg_launch_us']:7.2f} ms={r['ms_per_step']:.4f} frac={r['frac']}")
PY
tail -3 gpurun_out/r4c6_ab.err
