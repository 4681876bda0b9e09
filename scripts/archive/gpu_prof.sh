#!/bin/bash
# usage: bash scripts/gpu_prof.sh <tag>  -- kernel-trace stats + HBM traffic counters (separate passes) of a short bench run
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
T=${1:-p}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-graph"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_prof -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $R/gpurun_out/${T}_prof_stdout.log 2>&1
echo "stats rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${T}_pmc_fetch -o p -- $CMD > $R/gpurun_out/${T}_pmc_fetch.log 2>&1
echo "fetch rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${T}_pmc_write -o p -- $CMD > $R/gpurun_out/${T}_pmc_write.log 2>&1
echo "write rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/${T}_pmc_sq -o p -- $CMD > $R/gpurun_out/${T}_pmc_sq.log 2>&1
echo "sq rc=$?"
ls $R/gpurun_out/${T}_pmc_fetch $R/gpurun_out/${T}_prof
