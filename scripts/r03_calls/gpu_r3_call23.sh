#!/bin/bash
# band conv kernel: SQ counters (own PMC pass, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/c23_sq -o p -- python $R/scripts/bench_conv128.py 86 > /tmp/c23_sq.log 2>&1
echo "pmc sq rc=$?"
python $R/scripts/sq_counter_summary.py $(find /tmp/c23_sq -name "*counter_collection.csv" | head -1) $R/gpurun_out/r3c23_sq.json
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d /tmp/c23_lds -o p -- python $R/scripts/bench_conv128.py 86 > /tmp/c23_lds.log 2>&1
echo "pmc lds rc=$?"; tail -3 /tmp/c23_lds.log
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/c23_lds/**/*counter_collection.csv', recursive=True)
if f:
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name'][:60]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_INSTS_LDS': n[k] += 1
    for k, c in acc.items():
        if 'conv3x3' in k:
            print(k, n[k], {a: round(b / max(n[k], 1)) for a, b in c.items()})
PY
