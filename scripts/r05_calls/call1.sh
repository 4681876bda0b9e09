#!/bin/bash
# r05 call 1: the new px kernels + precise plan + dual x3 GEMM (tests), the changed runtime / determinism tests for the timed launch structure,
# A/B of the precise plan on FasterViT-4 (dual vs concatenated x3 GEMMs, 1 / 2 / 3 stream shards), kernel-trace stats of the precise plan
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r5c1_summary.log
: > $S
timeout 900 python -m pytest tests/test_gpu_px.py -q -m gpu -s -x > gpurun_out/r5c1_px.log 2>&1
echo "px tests rc=$?" >> $S; grep -E "max-abs|dual|precise|passed|failed|Error|error" gpurun_out/r5c1_px.log | cut -c1-250 | tail -60 >> $S
timeout 600 python -m pytest tests/test_gpu_x3.py -q -m gpu -x > gpurun_out/r5c1_x3.log 2>&1
echo "x3 tests rc=$?" >> $S; tail -3 gpurun_out/r5c1_x3.log >> $S
timeout 900 python -m pytest tests/test_gpu_runtime.py tests/test_gpu_determinism.py -q -m gpu -s -k "bench_configuration" > gpurun_out/r5c1_rt.log 2>&1
echo "runtime/determinism (timed structure) rc=$?" >> $S; grep -E "max-abs|passed|failed|Error" gpurun_out/r5c1_rt.log | cut -c1-220 | tail -12 >> $S
ab() {
  E=$1; shift
  env $E timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-modes --prof-steps 1 "$@" > gpurun_out/r5c1_ab.json 2>> gpurun_out/r5c1_ab.err
  python - "$E $*" <<'PY' >> gpurun_out/r5c1_summary.log
import json, sys
try:
    d = json.loads(open('gpurun_out/r5c1_ab.json').read().strip().splitlines()[-1])
    r = d.get('roofline') or {}
    print(f"{sys.argv[1][:150]:150s} {d['ms_per_step']:.3f} ms/step {d['value']:.1f} img/s err {d['parity']['logits_max_abs_err'] if d.get('parity') else None} dom {r.get('kernel')} {r.get('avg_launch_us')} us frac {r.get('frac')}")
except Exception as e:
    print(sys.argv[1][:150], "FAILED", e)
PY
}
F4="--model faster_vit_4_224 --batch 128 --join-from 0 --operand f16x3 --precise"
ab FVIT_TUNE_gemm_x3_dual=1 $F4 --streams 2
ab FVIT_TUNE_gemm_x3_dual=0 $F4 --streams 2
ab FVIT_TUNE_gemm_x3_dual=1 $F4 --streams 1
ab FVIT_TUNE_gemm_x3_dual=1 $F4 --streams 3
AR="--model faster_vit_4_any_res --batch 8 --input-size 576x960 --join-from 0 --operand f16x3 --precise"
KW="{'resolution':[576,960],'window_size':[7,7,12,6],'ct_size':2}"
ab FVIT_TUNE_gemm_x3_dual=1 $AR --model-kwargs "$KW" --streams 2
ab FVIT_TUNE_gemm_x3_dual=1 $AR --model-kwargs "$KW" --streams 1
ab X=1 --operand f16x3 --precise --streams 2 --join-from 3
tail -5 gpurun_out/r5c1_ab.err >> $S
cp gpurun_out/bench_detail.json gpurun_out/r5c1_last_detail.json
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/r5c1_stats -o p -- python $R/bench.py $F4 --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-modes --no-graph --prof-steps 0 > /tmp/r5c1_stats.log 2>&1
echo "precise FasterViT-4 stats rc=$?" >> $R/$S
DB=$(find /tmp/r5c1_stats -name "*.db" | head -1)
python $R/scripts/summarize_rocprof_db.py $DB $R/gpurun_out/r5c1_fvit4_precise_rocprof >> $R/$S 2>&1
cd $R
cat $S | cut -c1-400
