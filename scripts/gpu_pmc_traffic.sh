#!/bin/bash
# usage: bash scripts/gpu_pmc_traffic.sh <tag>  -- FETCH_SIZE / WRITE_SIZE passes of bench.py --no-graph, summarised on the box
R=$GRAFT_REPO_ROOT
T=${1:-pmc}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-graph"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/${T}_fetch -o p -- $CMD > /tmp/${T}_fetch.log 2>&1
echo "fetch rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/${T}_write -o p -- $CMD > /tmp/${T}_write.log 2>&1
echo "write rc=$?"
python $R/scripts/pmc_traffic_summary.py $(find /tmp/${T}_fetch -name "*counter_collection.csv" | head -1) $(find /tmp/${T}_write -name "*counter_collection.csv" | head -1) $R/gpurun_out/${T}_pmc_traffic.json | head -14
