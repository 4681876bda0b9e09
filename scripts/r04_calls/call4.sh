#!/bin/bash
# r04 call 4: generalized hat_backward (every 224 entrypoint geometry, propagation, padding, DropPath, train mode) + FasterViT-4 shard / join sweep
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_gpu_runtime.py -q -m gpu -s 2>&1 | grep -E "worst|relative L2|upstream|bf16 backward|passed|failed|Error|error|assert|FAILED|level" | cut -c1-260 | tail -60 > gpurun_out/r4c4_bwd.log; cat gpurun_out/r4c4_bwd.log
ab() {
  timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-modes --prof-steps 0 "$@" > gpurun_out/r4c4_ab.json 2>> gpurun_out/r4c4_ab.err
  python - "$*" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/r4c4_ab.json').read().strip().splitlines()[-1])
print(f"{sys.argv[1]:90s} {d['ms_per_step']:.4f} ms/step {d['value']:.1f} img/s")
PY
}
F4="--model faster_vit_4_224 --batch 128"
ab $F4 --streams 3 --join-from 0
ab $F4 --streams 3 --join-from 3
ab $F4 --streams 2 --join-from 3
ab $F4 --streams 2 --join-from 0
ab $F4 --streams 3 --join-from 0
AR="--model faster_vit_4_any_res --batch 8 --input-size 576x960"
KW="{'resolution':[576,960],'window_size':[7,7,12,6],'ct_size':2}"
ab $AR --model-kwargs "$KW" --streams 2 --join-from 0
ab $AR --model-kwargs "$KW" --streams 2 --join-from 3
tail -3 gpurun_out/r4c4_ab.err
