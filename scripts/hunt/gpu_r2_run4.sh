#!/bin/bash
# round 2, GPU call 4: repeatability stress (race hunt), LayerNorm-in-GEMM kernels: tests + end-to-end A/B
cd $GRAFT_REPO_ROOT
T=${1:-r2d}
mkdir -p gpurun_out
timeout 400 python scripts/race_hunt.py 30 > gpurun_out/${T}_race_hunt.log 2>&1
echo "race hunt rc=$?"; grep -v "0 of" gpurun_out/${T}_race_hunt.log | tail -40; grep -c "0 of" gpurun_out/${T}_race_hunt.log
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_determinism.py tests/test_head_train.py -q -x -m gpu -k "ln_gemm or knobs or repeatable or head or gemm_residual" > gpurun_out/${T}_test_k.log 2>&1
echo "kernel tests rc=$?"; tail -5 gpurun_out/${T}_test_k.log
bash scripts/gpu_sweep.sh ${T} "--steps 40" - "FVIT_TUNE_ln_gemm=0" "FVIT_TUNE_pe_preadd=0" - "FVIT_TUNE_ln_gemm=0" "FVIT_TUNE_lngemm_extra_wgs=300" > /dev/null 2>&1
cat gpurun_out/${T}_sweep.log
timeout 300 python bench.py --steps 30 --no-secondary --no-cpu-baseline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/'+"$T"+'_bench.json').read())
print(d['value'], d['ms_per_step'], d['launches_per_step'], d['kernel_ms_per_step_serialized'])
for r in d['roofline_shapes'][:26]:
    print("%-26s %5d %3d %8.2f %8.4f %7.1f %6.0f"%(r['kernel'][:26],r['workgroups'],r['launches_per_step'],r['avg_launch_us'],r['ms_per_step'],r['tflops'],r['gbs']))
PY
