#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for s in 1 2 3 4; do
  timeout 600 python bench.py --streams $s --no-cpu-baseline --prof-steps 1 > gpurun_out/bench_streams$s.json 2> gpurun_out/bench_streams$s.err
  echo "streams=$s rc=$?"; python -c "import json; d=json.load(open('gpurun_out/bench_streams$s.json')); print(d['value'], d['ms_per_step'], d['config']['launch'])"
done
