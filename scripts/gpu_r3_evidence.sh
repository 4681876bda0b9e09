#!/bin/bash
# round-3 evidence run (through gpurun): full GPU test suite, default bench line, rocprofv3 kernel-trace stats of the same command,
# HBM-traffic PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs) and one SQ-counter pass.  usage: bash scripts/gpu_r3_evidence.sh <tag> [notest]
# (the secondary configurations' stats / PMC passes: scripts/r03_calls/gpu_r3_call7.sh / gpu_r3_call8.sh)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
T=${1:-r3e}
mkdir -p gpurun_out
S=gpurun_out/${T}_summary.log
: > $S
if [ "$2" != "notest" ]; then
  timeout 1500 python -m pytest tests -q -m gpu -s > gpurun_out/${T}_test_gpu.log 2>&1
  echo "pytest-gpu rc=$?" >> $S
  tail -3 gpurun_out/${T}_test_gpu.log >> $S
  grep -h "err \|rel err\|differs" gpurun_out/${T}_test_gpu.log >> $S
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> $S 2>&1
fi
timeout 900 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc=$?" >> $S
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_prof -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-modes > $R/gpurun_out/${T}_prof_stdout.log 2>&1
echo "rocprof stats rc=$?" >> $R/$S
DB=$(find $R/gpurun_out/${T}_prof -name "*.db" | head -1)
python $R/scripts/summarize_rocprof_db.py $DB $R/gpurun_out/${T}_rocprof >> $R/$S 2>&1
CMD="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-modes --no-graph --prof-steps 0"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/${T}_fetch -o p -- $CMD > /tmp/${T}_fetch.log 2>&1
echo "pmc fetch rc=$?" >> $R/$S
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/${T}_write -o p -- $CMD > /tmp/${T}_write.log 2>&1
echo "pmc write rc=$?" >> $R/$S
python $R/scripts/pmc_traffic_summary.py $(find /tmp/${T}_fetch -name "*counter_collection.csv" | head -1) $(find /tmp/${T}_write -name "*counter_collection.csv" | head -1) $R/gpurun_out/${T}_pmc_traffic.json >> $R/$S 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/${T}_sq -o p -- $CMD > /tmp/${T}_sq.log 2>&1
echo "pmc sq rc=$?" >> $R/$S
python $R/scripts/sq_counter_summary.py $(find /tmp/${T}_sq -name "*counter_collection.csv" | head -1) $R/gpurun_out/${T}_sq_counters.json >> $R/$S 2>&1
rm -rf $R/gpurun_out/${T}_prof
cd $R
cat $S | cut -c1-400
head -c 1200 gpurun_out/${T}_bench.json
