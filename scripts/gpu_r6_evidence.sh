#!/bin/bash
# round-6 evidence run (through gpurun): full GPU test suite + smoke, the driver-form bench line (+ full record), rocprofv3 kernel-trace stats of the same
# command, HBM-traffic PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs) and one SQ-counter pass of the headline configuration; kernel-trace stats of
# the two secondary configurations in their TIMED (precise) form and of FasterViT-4 in the fast form; PMC traffic of FasterViT-4 precise.
# usage: bash scripts/gpu_r6_evidence.sh <tag> [notest]
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
T=${1:-r6e}
mkdir -p gpurun_out
S=gpurun_out/${T}_summary.log
: > $S
if [ "$2" != "notest" ]; then
  timeout 2400 python -m pytest tests -q -m gpu -s > gpurun_out/${T}_test_gpu.log 2>&1
  echo "pytest-gpu rc=$?" >> $S
  tail -3 gpurun_out/${T}_test_gpu.log >> $S
  grep -h "err \|rel err\|differs\|worst\|relative L2\|max-abs\|precise" gpurun_out/${T}_test_gpu.log | cut -c1-220 >> $S
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> $S 2>&1
fi
( time timeout 900 python bench.py --record gpurun_out/${T}_bench_detail.json ) > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc=$? line bytes $(tail -1 gpurun_out/${T}_bench.json | wc -c)" >> $S
tail -4 gpurun_out/${T}_bench.err >> $S
# the driver's launch form for N > 1 (one process per GPU under torch.distributed.run, RCCL group), here with the one GPU this box has
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-modes --no-train-step --prof-steps 0 > gpurun_out/${T}_torchrun1.log 2>&1
echo "torchrun world-1 rc=$? $(grep '^{' gpurun_out/${T}_torchrun1.log | tail -1 | cut -c1-160)" >> $S
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_prof -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-modes --no-train-step > $R/gpurun_out/${T}_prof_stdout.log 2>&1
echo "rocprof stats rc=$?" >> $R/$S
DB=$(find $R/gpurun_out/${T}_prof -name "*.db" | head -1)
python $R/scripts/summarize_rocprof_db.py $DB $R/gpurun_out/${T}_rocprof >> $R/$S 2>&1
CMD="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-modes --no-train-step --no-graph --prof-steps 0"
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
  timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/${T}_${N}_stats -o p -- python $R/bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-modes --no-train-step --no-graph --prof-steps 0 > /tmp/${T}_${N}_stats.log 2>&1
  echo "$N stats rc=$?" >> $R/$S
  DB=$(find /tmp/${T}_${N}_stats -name "*.db" | head -1)
  python $R/scripts/summarize_rocprof_db.py $DB $R/gpurun_out/${T}_${N}_rocprof >> $R/$S 2>&1
}
KW="{'resolution':[576,960],'window_size':[7,7,12,6],'ct_size':2}"
run_cfg faster_vit_4_224_precise --model faster_vit_4_224 --batch 128 --streams 1 --join-from 0 --operand f16x3 --precise
run_cfg faster_vit_4_any_res_precise --model faster_vit_4_any_res --batch 8 --input-size 576x960 --model-kwargs "$KW" --streams 1 --join-from 0 --operand f16x3 --precise
run_cfg faster_vit_4_224_fast --model faster_vit_4_224 --batch 128 --streams 1 --join-from 0
CMD4="python $R/bench.py --model faster_vit_4_224 --batch 128 --streams 1 --join-from 0 --operand f16x3 --precise --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-modes --no-train-step --no-graph --prof-steps 0"
timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/${T}_f4fetch -o p -- $CMD4 > /tmp/${T}_f4fetch.log 2>&1
echo "pmc fetch (FasterViT-4 precise) rc=$?" >> $R/$S
timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/${T}_f4write -o p -- $CMD4 > /tmp/${T}_f4write.log 2>&1
echo "pmc write (FasterViT-4 precise) rc=$?" >> $R/$S
python $R/scripts/pmc_traffic_summary.py $(find /tmp/${T}_f4fetch -name "*counter_collection.csv" | head -1) $(find /tmp/${T}_f4write -name "*counter_collection.csv" | head -1) $R/gpurun_out/${T}_pmc_traffic_faster_vit_4_224.json >> $R/$S 2>&1
cd $R
ls gpurun_out | grep ${T} >> $S
cat $S | cut -c1-330
