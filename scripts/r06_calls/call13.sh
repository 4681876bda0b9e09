#!/bin/bash
# r06 call 13: the new default (2 whole-batch steps in flight): runtime tests, the driver-form bench line, kernel-trace stats of the timed region
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r6c13_summary.log
: > $S
timeout 900 python -m pytest tests/test_gpu_runtime.py tests/test_gpu_determinism.py -q -m gpu -s -x > gpurun_out/r6c13_tests.log 2>&1
echo "tests rc=$?" >> $S; tail -4 gpurun_out/r6c13_tests.log | cut -c1-300 >> $S; grep -h "pipelined" gpurun_out/r6c13_tests.log | cut -c1-200 >> $S
( time timeout 1200 python bench.py --record gpurun_out/r6c13_bench_detail.json ) > gpurun_out/r6c13_bench.json 2> gpurun_out/r6c13_bench.err
echo "bench rc=$? line bytes $(tail -1 gpurun_out/r6c13_bench.json | wc -c)" >> $S
tail -4 gpurun_out/r6c13_bench.err >> $S
python - <<'PY' >> $S
import json
d = json.loads(open('gpurun_out/r6c13_bench.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value', 'ms_per_step', 'parity', 'bf16_under_1e-3_images_per_s')})
print(d['config']['launch'])
print(d.get('roofline'))
for s in d.get('secondary', []):
    print({k: s.get(k) for k in ('workload', 'value', 'ms_per_step', 'launch')}, (s.get('parity') or {}).get('logits_max_abs_err'), (s.get('fast') or {}).get('value'))
for k in d:
    if k.startswith('parity_'):
        print(k, d[k])
print(d.get('train_step'))
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r6c13_prof -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-modes --no-train-step > $R/gpurun_out/r6c13_prof_stdout.log 2>&1
echo "rocprof stats rc=$?" >> $R/$S
DB=$(find $R/gpurun_out/r6c13_prof -name "*.db" | head -1)
python $R/scripts/summarize_rocprof_db.py $DB $R/gpurun_out/r6c13_rocprof >> $R/$S 2>&1
rm -rf $R/gpurun_out/r6c13_prof
cd $R
cat $S | cut -c1-600
