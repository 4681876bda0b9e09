#!/bin/bash
# round-4 evidence run (through gpurun): full GPU test suite + smoke, default bench line (2 stream shards joined in front of level 3), rocprofv3
# kernel-trace stats of the same command, HBM-traffic PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs) and one SQ-counter pass; kernel-trace stats of
# the two secondary configurations (fast mode) and of FasterViT-4 in the precise mode (module mode + f16x3); a small shard-count sweep.
# usage: bash scripts/gpu_r4_evidence.sh <tag> [notest]
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
T=${1:-r4e}
mkdir -p gpurun_out
S=gpurun_out/${T}_summary.log
: > $S
if [ "$2" != "notest" ]; then
  timeout 2400 python -m pytest tests -q -m gpu -s > gpurun_out/${T}_test_gpu.log 2>&1
  echo "pytest-gpu rc=$?" >> $S
  tail -3 gpurun_out/${T}_test_gpu.log >> $S
  grep -h "err \|rel err\|differs\|worst\|relative L2" gpurun_out/${T}_test_gpu.log | cut -c1-220 >> $S
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> $S 2>&1
fi
( time timeout 900 python bench.py ) > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc=$? line bytes $(tail -1 gpurun_out/${T}_bench.json | wc -c)" >> $S
tail -4 gpurun_out/${T}_bench.err >> $S
cp gpurun_out/bench_detail.json gpurun_out/${T}_bench_detail.json
for a in "" "--streams 3 --join-from 0" "--streams 1 --join-from 0" "--streams 2 --join-from 2" "--streams 3 --join-from 3"; do
  timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary --no-modes --prof-steps 0 $a > gpurun_out/${T}_tmp.json 2>> gpurun_out/${T}.err
  echo "sweep [$a]: $(python -c "import json;d=json.loads(open('gpurun_out/${T}_tmp.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], 'ms/step', d['value'], 'img/s')")" >> $S
done
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
run_cfg() {   # name, bench args: kernel-trace stats only
  N=$1; shift
  timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/${T}_${N}_stats -o p -- python $R/bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-modes --no-graph --prof-steps 0 > /tmp/${T}_${N}_stats.log 2>&1
  echo "$N stats rc=$?" >> $R/$S
  DB=$(find /tmp/${T}_${N}_stats -name "*.db" | head -1)
  python $R/scripts/summarize_rocprof_db.py $DB $R/gpurun_out/${T}_${N}_rocprof >> $R/$S 2>&1
}
run_cfg faster_vit_4_224 --model faster_vit_4_224 --batch 128 --streams 3 --join-from 0
run_cfg faster_vit_4_any_res --model faster_vit_4_any_res --batch 8 --input-size 576x960 --model-kwargs "{'resolution':[576,960],'window_size':[7,7,12,6],'ct_size':2}" --streams 2 --join-from 0
run_cfg faster_vit_4_224_precise_f16x3 --model faster_vit_4_224 --batch 128 --mode module --conv-dtype f32 --operand f16x3
cd $R
ls gpurun_out | grep ${T} >> $S
cat $S | cut -c1-330
