#!/bin/bash
cd $GRAFT_REPO_ROOT
T=${1:-r4f}
mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/${T}_smoke.log
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > gpurun_out/${T}_torchrun_bench.json 2> gpurun_out/${T}_torchrun_bench.err
echo "torchrun bench rc=$?"; python - <<PY
import json
d = json.loads(open("gpurun_out/${T}_torchrun_bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "n_gpus", "ms_per_step", "steps", "warmup")}, d["roofline"]["kernel"], d["roofline"]["frac"])
PY
WORLD_SIZE=2 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29518 timeout 60 python -c "
import os
from fastervit_amd import dp
print('env_world', dp.env_world())
" 2>&1 | tail -1
