#!/bin/bash
# full GPU test suite + repeatability hunt + short bench
cd $GRAFT_REPO_ROOT
T=${1:-r3h}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/${T}_pytest_gpu.log
timeout 400 python scripts/race_hunt4.py 12 > gpurun_out/${T}_race_hunt4.log 2>&1
echo "rc=$?"; grep -v "amdgpu.ids\|UserWarning\|stage_forward(" gpurun_out/${T}_race_hunt4.log | tail -6
timeout 600 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc=$?"; python - <<PY
import json
d = json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "parity", "parity_bf16") if k in d})
PY
