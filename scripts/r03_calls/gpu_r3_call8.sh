#!/bin/bash
# r03 call 8: GEMM knob sweep on faster_vit_4_224 (batch 128, 3 stream shards), then the any-res PMC / stats passes
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
T=r3c8
mkdir -p gpurun_out
S=$R/gpurun_out/${T}_summary.log
: > $S
for kn in "" "FVIT_TUNE_gemm_ring=3" "FVIT_TUNE_gemm_ring=4" "FVIT_TUNE_gemm_bm64_max_grid=0" "FVIT_TUNE_gemm_bm64_max_grid=250" "FVIT_TUNE_gemm256_min_tiles=96" "FVIT_TUNE_gemm_stagger=1" "FVIT_TUNE_gemm_3stage_max_grid=600 FVIT_TUNE_gemm_bm64_max_grid=0"; do
  env $kn timeout 300 python bench.py --model faster_vit_4_224 --batch 128 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-modes --prof-steps 0 > gpurun_out/${T}_tmp.json 2>> gpurun_out/${T}.err
  echo "fv4 [$kn]: $(python -c "import json;d=json.load(open('gpurun_out/${T}_tmp.json'));print(d['ms_per_step'], 'ms/step', d['value'], 'img/s')")" >> $S
done
cd /tmp && export TMPDIR=/tmp
N=faster_vit_4_any_res
CMD="python $R/bench.py --model faster_vit_4_any_res --batch 8 --input-size 576x960 --model-kwargs {'resolution':[576,960],'window_size':[7,7,12,6],'ct_size':2} --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-modes --no-graph --prof-steps 0"
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/${T}_${N}_stats -o p -- $CMD > /tmp/${T}_${N}_stats.log 2>&1
echo "$N stats rc=$?" >> $S
tail -3 /tmp/${T}_${N}_stats.log >> $S
DB=$(find /tmp/${T}_${N}_stats -name "*.db" | head -1)
python $R/scripts/summarize_rocprof_db.py $DB $R/gpurun_out/${T}_${N}_rocprof >> $S 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/${T}_${N}_fetch -o p -- $CMD > /tmp/${T}_${N}_fetch.log 2>&1
echo "$N fetch rc=$?" >> $S
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/${T}_${N}_write -o p -- $CMD > /tmp/${T}_${N}_write.log 2>&1
echo "$N write rc=$?" >> $S
python $R/scripts/pmc_traffic_summary.py $(find /tmp/${T}_${N}_fetch -name "*counter_collection.csv" | head -1) $(find /tmp/${T}_${N}_write -name "*counter_collection.csv" | head -1) $R/gpurun_out/${T}_pmc_${N}.json >> $S 2>&1
cat $S | cut -c1-300
