#!/bin/bash
# r06 call 28: the final code (two-term long-window attention added after the evidence run r6e5, no timed kernel touched): full GPU suite, smoke, the driver-form bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r6c28_summary.log
: > $S
timeout 2400 python -m pytest tests -q -m gpu -s > gpurun_out/r6c28_test_gpu.log 2>&1
echo "pytest-gpu rc=$?" >> $S
grep -E "passed|failed" gpurun_out/r6c28_test_gpu.log | tail -2 >> $S
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> $S 2>&1
( time timeout 900 python bench.py --record gpurun_out/r6c28_bench_detail.json ) > gpurun_out/r6c28_bench.json 2> gpurun_out/r6c28_bench.err
echo "bench rc=$? line bytes $(tail -1 gpurun_out/r6c28_bench.json | wc -c)" >> $S
tail -4 gpurun_out/r6c28_bench.err >> $S
python - <<'PY' >> $S
import json
d = json.loads(open('gpurun_out/r6c28_bench.json').read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step'], d['parity']['logits_max_abs_err'], d['parity']['images'], 'frac', d['roofline']['frac'])
for s in d.get('secondary', []):
    print(s['workload'][:44], s['value'], s['parity']['logits_max_abs_err'], (s.get('fast') or {}).get('value'))
PY
cat $S | cut -c1-250
