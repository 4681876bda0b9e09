#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=gpurun_out/r3c16_summary.log
: > $S
for kn in "" "FVIT_TUNE_ct_variant=1" "FVIT_TUNE_stem_fused_grid=768" "FVIT_TUNE_stem_fused_grid=1024" "" "FVIT_TUNE_ct_variant=1" "FVIT_TUNE_stem_fused_grid=384" "FVIT_TUNE_attn_fused_min_rows=1 FVIT_TUNE_mlp_fused_min_rows=1"; do
  env $kn timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary --no-modes --prof-steps 0 > gpurun_out/r3c16_tmp.json 2>> gpurun_out/r3c16.err
  echo "[$kn]: $(python -c "import json;d=json.load(open('gpurun_out/r3c16_tmp.json'));print(d['ms_per_step'], 'ms/step', d['value'], 'img/s')")" >> $S
done
cat $S
