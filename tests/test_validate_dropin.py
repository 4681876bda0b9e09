"""BASELINE configs[0]: the reference's own validate.py, unmodified, driving our model (plumbing, no GPU).

Needs /root/reference (build container only).  The HAT stages have no CPU path in the product, so THIS TEST
substitutes the CPU oracle for the stage executor -- test infrastructure standing in for the GPU, never shipped."""
import json
import os
import subprocess
import sys

import pytest
import torch

REF = "/root/reference/fastervit/validate.py"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r'''
import sys, torch
sys.path.insert(0, %(root)r)
from fastervit_amd import hat_runtime
from oracle import hat_reference as hr

def oracle_stage(layer, x):                      # CPU stand-in for fvit_hat_stage_forward (test only)
    sd = {k: v for k, v in layer.state_dict().items()}
    blk = layer.blocks[0]
    return hr.hat_stage(x.float(), sd, "", depth=len(layer.blocks), heads=blk.attn.num_heads, ws=layer.window_size,
                        cw=blk.cr_window, input_resolution=list(x.shape[2:]), only_local=not blk.do_sr_hat,
                        do_propagation=blk.do_propagation, any_res=layer.any_res).to(x.dtype)

hat_runtime.stage_forward = oracle_stage
sys.path.insert(0, %(root)r + "/scripts")
import run_reference_validate
run_reference_validate.main(sys.argv[1:])
'''


@pytest.mark.skipif(not os.path.isfile(REF), reason="reference checkout not present (GPU box)")
def test_reference_validate_py_runs_unmodified(tmp_path):
    import fastervit_amd
    from tests.synth import synth_state_dict
    m = fastervit_amd.create_model("faster_vit_0_224")
    ck = str(tmp_path / "synthetic.pth.tar")
    torch.save({"state_dict": synth_state_dict(m.state_dict(), 7, "init")}, ck)
    drv = tmp_path / "drv.py"
    drv.write_text(DRIVER % {"root": ROOT})
    res_file = str(tmp_path / "res.json")
    cmd = [sys.executable, str(drv), REF, "--model", "faster_vit_0_224", "--checkpoint", ck, "-b", "8", "--device", "cpu",
           "--results-file", res_file, "--results-format", "json", "--workers", "0", "--no-prefetcher"]
    out = subprocess.run(cmd, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    res = json.load(open(res_file))
    res = res[0] if isinstance(res, list) else res
    assert res["model"] == "faster_vit_0_224"
    assert abs(float(res["param_count"]) - 31.4) < 0.01
    assert 0.0 <= float(res["top1"]) <= 100.0
