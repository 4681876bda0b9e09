#!/bin/bash
# r06 call 5: the whole GPU suite on the pruned library + the default bench.py run (driver form: no flags) with every leg
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r6c5_summary.log
: > $S
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r6c5_tests.log 2>&1
echo "tests rc=$?" >> $S; tail -12 gpurun_out/r6c5_tests.log | cut -c1-300 >> $S
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6c5_smoke.log 2>&1; echo "smoke rc=$?" >> $S; tail -2 gpurun_out/r6c5_smoke.log >> $S
( time timeout 1500 python bench.py ) > gpurun_out/r6c5_bench.json 2> gpurun_out/r6c5_bench.err
echo "bench rc=$?" >> $S; tail -4 gpurun_out/r6c5_bench.err >> $S
tail -1 gpurun_out/r6c5_bench.json | cut -c1-6000 >> $S
cp gpurun_out/bench_detail.json gpurun_out/r6c5_bench_detail.json
cat $S | cut -c1-6000
