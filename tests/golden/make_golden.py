#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference (/root/reference) in the build container.

    python tests/golden/make_golden.py            # all cases
    python tests/golden/make_golden.py tiny_hier  # one case

The reference is imported unmodified through the test-only timm shim (tests/golden/_shim); weights
and inputs are the deterministic synthetic ones of tests/synth.py, so nothing but the outputs needs
to be stored.  Outputs: tests/golden/<case>.npz (fp32) and tests/golden/state_keys.json (key/shape
digests of every entrypoint's state_dict).  /root/reference does not exist on the GPU box: tests
only ever read the committed files.
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, "_shim"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

from fastervit.models import create_model as ref_create_model  # noqa: E402
from fastervit.models.registry import list_models as ref_list_models  # noqa: E402

from tests.cases import CASES, SEED  # noqa: E402
from tests.synth import synth_input, synth_state_dict  # noqa: E402


def digest(sd):
    lines = sorted(f"{k}:{tuple(v.shape)}:{str(v.dtype).replace('torch.', '')}" for k, v in sd.items())
    return hashlib.sha256("\n".join(lines).encode()).hexdigest(), len(lines)


def run_case(name, case):
    torch.manual_seed(0)
    model = ref_create_model(case["entry"], **case["kwargs"]).eval()
    sd = synth_state_dict(model.state_dict(), SEED, case["family"])
    model.load_state_dict(sd, strict=True)
    x = synth_input(case["batch"], case["hw"][0], case["hw"][1], SEED)
    store = {}
    hooks = []
    for li in (2, 3):
        lvl = model.levels[li]
        hooks.append(lvl.register_forward_pre_hook(
            lambda m, inp, li=li: store.__setitem__(f"level{li}_in", inp[0].detach().clone())))
        if lvl.downsample is not None:
            hooks.append(lvl.downsample.register_forward_pre_hook(
                lambda m, inp, li=li: store.__setitem__(f"level{li}_out", inp[0].detach().clone())))
        else:
            hooks.append(lvl.register_forward_hook(
                lambda m, inp, out, li=li: store.__setitem__(f"level{li}_out", out.detach().clone())))
        if case["per_block"]:
            for bi, blk in enumerate(lvl.blocks):
                def hk(m, inp, out, li=li, bi=bi):
                    store[f"l{li}b{bi}_x"] = out[0].detach().clone()
                    if out[1] is not None:
                        store[f"l{li}b{bi}_ct"] = out[1].detach().clone()
                hooks.append(blk.register_forward_hook(hk))
            if hasattr(lvl, "global_tokenizer"):
                hooks.append(lvl.global_tokenizer.register_forward_hook(
                    lambda m, inp, out, li=li: store.__setitem__(f"l{li}_ct0", out.detach().clone())))
    with torch.no_grad():
        logits = model(x)
    for h in hooks:
        h.remove()
    out = {"logits": logits.numpy().astype(np.float32)}
    for k, v in store.items():
        a = v.numpy().astype(np.float32)
        if not case["per_block"] and k.startswith("level"):
            if not case.get("stage_maps", False):
                continue  # big models: logits only
            a = a[:1]  # full-size cases: keep image 0 of the stage inputs/outputs only
        out[k] = a
    d, n = digest(model.state_dict())
    out["state_digest"] = np.array([d])
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: logits {tuple(logits.shape)} |max| {logits.abs().max():.4f}  keys {n}  -> {os.path.getsize(path) / 1e6:.2f} MB")


def all_state_digests():
    res = {}
    orig_linspace = torch.linspace
    # the reference calls linspace(...).item() while building (FV:901), which a meta device cannot do
    torch.linspace = lambda *a, **k: orig_linspace(*a, **{**k, "device": "cpu"})
    for name in ref_list_models():
        with torch.device("meta"):
            m = ref_create_model(name)
        d, n = digest(m.state_dict())
        params = sum(p.numel() for p in m.parameters())
        res[name] = dict(sha256=d, keys=n, params=params)
        print(f"  {name}: {n} keys, {params / 1e6:.2f} M params")
    with open(os.path.join(HERE, "state_keys.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES)
    for n in names:
        if n == "state_keys":
            all_state_digests()
        else:
            run_case(n, CASES[n])
    if not sys.argv[1:]:
        all_state_digests()
