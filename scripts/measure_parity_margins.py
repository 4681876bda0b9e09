#!/usr/bin/env python3
"""Measured error of every assertion in tests/test_gpu_parity.py that has a relative tolerance (per-block goldens, tiny models, stress logits, smoke):
prints max error per test so that the asserted bounds can be kept at ~2x the measurement (VERDICT r05 'What's weak' 1e).  GPU only."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import hat_reference as hr   # noqa: E402  (a measurement tool, like the tests: not product code)
from tests.cases import CASES   # noqa: E402
from tests.util import build_product_model, case_input, load_golden, rel_err   # noqa: E402

TINY = [n for n, c in CASES.items() if c["per_block"]]
worst_blk = worst_ct = worst_lvl = worst_log = 0.0
for name in TINY:
    g = load_golden(name)
    model, _ = build_product_model(name, "cuda")
    bx = bc = 0.0
    for li in (2, 3):
        lvl = model.levels[li]
        ws = lvl.window_size
        xin = torch.from_numpy(g[f"level{li}_in"])
        H, W = xin.shape[2:]
        pad_b, pad_r = (ws - H % ws) % ws, (ws - W % ws) % ws
        xw = hr.window_partition(torch.nn.functional.pad(xin, (0, pad_r, 0, pad_b)), ws)
        ct = torch.from_numpy(g[f"l{li}_ct0"]) if f"l{li}_ct0" in g and lvl.blocks[0].do_sr_hat else None
        for bi, blk in enumerate(lvl.blocks):
            with torch.no_grad():
                xo, cto = blk(xw.cuda(), None if ct is None else ct.cuda())
            bx = max(bx, rel_err(xo.cpu(), g[f"l{li}b{bi}_x"]))
            if ct is not None:
                bc = max(bc, rel_err(cto.cpu(), g[f"l{li}b{bi}_ct"]))
                ct = torch.from_numpy(g[f"l{li}b{bi}_ct"])
            xw = torch.from_numpy(g[f"l{li}b{bi}_x"])
    x = case_input(name).cuda()
    feats = {}
    hooks = []
    for li in (2, 3):
        lvl = model.levels[li]
        if lvl.downsample is not None:
            hooks.append(lvl.downsample.register_forward_pre_hook(lambda m, inp, li=li: feats.__setitem__(li, inp[0].float().cpu())))
        else:
            hooks.append(lvl.register_forward_hook(lambda m, inp, out, li=li: feats.__setitem__(li, out.float().cpu())))
    with torch.no_grad():
        logits = model(x).float().cpu()
    el = max(rel_err(feats[li], g[f"level{li}_out"]) for li in (2, 3))
    eg = rel_err(logits, g["logits"])
    print(f"{name:28s} per-block x {bx:.3e} ct {bc:.3e} | level out {el:.3e} logits {eg:.3e}")
    worst_blk, worst_ct, worst_lvl, worst_log = max(worst_blk, bx), max(worst_ct, bc), max(worst_lvl, el), max(worst_log, eg)
print(f"WORST per-block x {worst_blk:.3e} ct {worst_ct:.3e} level {worst_lvl:.3e} logits {worst_log:.3e}")
g = load_golden("fvit0_224_stress")
model, _ = build_product_model("fvit0_224_stress", "cuda")
with torch.no_grad():
    logits = model(case_input("fvit0_224_stress").cuda()).float().cpu()
print(f"fvit0_224_stress logits rel {rel_err(logits, g['logits']):.3e}")
