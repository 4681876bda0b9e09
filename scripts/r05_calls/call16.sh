#!/bin/bash
# r05 call 16: SQ counters of the precise FasterViT-4 plan (eager, 2 stream shards): MFMA busy / issue stall / wait of the x3 GEMMs and the two-term convs
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD4="python $R/bench.py --model faster_vit_4_224 --batch 128 --streams 2 --join-from 0 --operand f16x3 --precise --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-modes --no-train-step --no-graph --prof-steps 0"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/r5c16_sq -o p -- $CMD4 > /tmp/r5c16_sq.log 2>&1
echo "pmc sq rc=$?"
python $R/scripts/sq_counter_summary.py $(find /tmp/r5c16_sq -name "*counter_collection.csv" | head -1) $R/gpurun_out/r5c16_sq_counters_faster_vit_4_224_precise.json
