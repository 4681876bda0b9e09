#!/bin/bash
cd $GRAFT_REPO_ROOT
T=${1:-r3u}
mkdir -p gpurun_out
timeout 120 ./scripts/probes/regstream_probe > gpurun_out/${T}_regstream_probe.log 2>&1
echo "rc=$?"; cat gpurun_out/${T}_regstream_probe.log
