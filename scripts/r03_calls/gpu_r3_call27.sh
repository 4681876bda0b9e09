#!/bin/bash
# rocprofv3 stats + HBM-traffic PMC passes of the two secondary configurations after the ping-pong GEMM tile became the default
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
T=r3c27
mkdir -p gpurun_out
S=$R/gpurun_out/${T}_summary.log
: > $S
cd /tmp && export TMPDIR=/tmp
run_cfg() {   # name, bench args
  N=$1; shift
  CMD="python $R/bench.py $* --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-modes --no-graph --prof-steps 0"
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/${T}_${N}_stats -o p -- $CMD > /tmp/${T}_${N}_stats.log 2>&1
  echo "$N stats rc=$?" >> $S
  DB=$(find /tmp/${T}_${N}_stats -name "*.db" | head -1)
  python $R/scripts/summarize_rocprof_db.py $DB $R/gpurun_out/${T}_${N}_rocprof >> $S 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/${T}_${N}_fetch -o p -- $CMD > /tmp/${T}_${N}_fetch.log 2>&1
  echo "$N fetch rc=$?" >> $S
  timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/${T}_${N}_write -o p -- $CMD > /tmp/${T}_${N}_write.log 2>&1
  echo "$N write rc=$?" >> $S
  python $R/scripts/pmc_traffic_summary.py $(find /tmp/${T}_${N}_fetch -name "*counter_collection.csv" | head -1) $(find /tmp/${T}_${N}_write -name "*counter_collection.csv" | head -1) $R/gpurun_out/${T}_pmc_${N}.json >> $S 2>&1
}
run_cfg faster_vit_4_224 --model faster_vit_4_224 --batch 128
run_cfg faster_vit_4_any_res --model faster_vit_4_any_res --batch 8 --streams 2 --input-size 576x960 --model-kwargs "{'resolution':[576,960],'window_size':[7,7,12,6],'ct_size':2}"
ls $R/gpurun_out | grep ${T} >> $S
cat $S | cut -c1-300
