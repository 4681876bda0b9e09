#!/bin/bash
cd $GRAFT_REPO_ROOT
T=${1:-r4k}
mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc=$?"; head -c 600 gpurun_out/${T}_bench.json
