"""scripts/run_sharded_validate.py (the per-process replacement of validate.py's nn.DataParallel path): two gloo ranks on the CPU
evaluate disjoint halves of the same sample stream and agree with one process on the whole stream.  The HAT stages have no CPU
path in the product, so THIS TEST substitutes the CPU oracle for the stage executor (test infrastructure, as in
test_validate_dropin.py); on the GPU the same script runs the hipGraph runner (tests/test_gpu_runtime.py covers that path)."""
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r'''
import sys, torch
sys.path.insert(0, %(root)r)
from fastervit_amd import hat_runtime
from oracle import hat_reference as hr

def oracle_stage(layer, x, tokenizer=None, out=None):     # CPU stand-in for fvit_hat_stage_forward (test only)
    sd = {k: v for k, v in layer.state_dict().items()}
    blk = layer.blocks[0]
    return hr.hat_stage(x.float(), sd, "", depth=len(layer.blocks), heads=blk.attn.num_heads, ws=layer.window_size,
                        cw=blk.cr_window, input_resolution=list(x.shape[2:]), only_local=not blk.do_sr_hat,
                        do_propagation=blk.do_propagation, any_res=layer.any_res).to(x.dtype)

hat_runtime.stage_forward = oracle_stage
sys.path.insert(0, %(root)r + "/scripts")
import run_sharded_validate
run_sharded_validate.main(sys.argv[1:])
'''

KW = "{'depths': [1, 1, 2, 1], 'num_heads': [1, 1, 2, 4], 'dim': 16, 'in_dim': 16, 'num_classes': 10}"


def _run(tmp_path, world, tag):
    drv = tmp_path / "drv.py"
    drv.write_text(DRIVER % {"root": ROOT})
    logits = str(tmp_path / f"logits_{tag}.pt")
    res = str(tmp_path / f"res_{tag}.json")
    common = [str(drv), "--model", "faster_vit_0_224", "--model-kwargs", KW, "-b", "3", "--synthetic", "10", "--device", "cpu",
              "--gather-logits", logits, "--results-file", res]
    if world == 1:
        cmd = [sys.executable] + common
    else:
        port = 29600 + (os.getpid() % 300)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + common
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run(cmd, cwd=str(tmp_path), capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    return json.load(open(res)), torch.load(logits)


def test_two_ranks_equal_one_process(tmp_path):
    r1, l1 = _run(tmp_path, 1, "w1")
    r2, l2 = _run(tmp_path, 2, "w2")
    assert r1["samples"] == r2["samples"] == 10 and r2["world_size"] == 2
    assert (r1["top1"], r1["top5"]) == (r2["top1"], r2["top5"])
    assert l1.shape == l2.shape == (10, 10)
    assert (l1 - l2).abs().max().item() < 1e-5      # same sample -> same logits whichever rank evaluated it
