#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r3c10_summary.log
: > $S
for st in 3 2 1; do
  timeout 300 python bench.py --model faster_vit_4_224 --batch 128 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-modes --prof-steps 0 --streams $st > gpurun_out/r3c10_tmp.json 2>> gpurun_out/r3c10.err
  echo "fv4 streams=$st: $(python -c "import json;d=json.load(open('gpurun_out/r3c10_tmp.json'));print(d['ms_per_step'], 'ms/step', d['value'], 'img/s')")" >> $S
done
for st in 3 2 1; do
  timeout 300 python bench.py --model faster_vit_4_any_res --batch 8 --input-size 576x960 --model-kwargs "{'resolution':[576,960],'window_size':[7,7,12,6],'ct_size':2}" --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-modes --prof-steps 0 --streams $st > gpurun_out/r3c10_tmp.json 2>> gpurun_out/r3c10.err
  echo "anyres streams=$st: $(python -c "import json;d=json.load(open('gpurun_out/r3c10_tmp.json'));print(d['ms_per_step'], 'ms/step', d['value'], 'img/s')")" >> $S
done
cat $S; tail -3 gpurun_out/r3c10.err
