"""Pins the CPU oracle (oracle/) against golden vectors produced by the real reference
(tests/golden/make_golden.py).  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from oracle import hat_reference as hr
from oracle.model_reference import model_forward
from tests.cases import CASES
from tests.util import GOLDEN_DIR, build_product_model, case_input, load_golden, max_abs

TINY = [n for n, c in CASES.items() if c["per_block"]]
FULL_FAST = ["fvit0_224", "fvit0_224_stress"]
FULL_SLOW = ["fvit4_224", "fvit4_anyres_576x960", "fvit4_21k_384"]


def _digest(sd):
    lines = sorted(f"{k}:{tuple(v.shape)}:{str(v.dtype).replace('torch.', '')}" for k, v in sd.items())
    return hashlib.sha256("\n".join(lines).encode()).hexdigest()


@pytest.mark.parametrize("name", TINY)
def test_oracle_matches_reference_per_block(name):
    """Oracle (fp32) vs reference: logits, stage outputs and every HAT block's (x, ct)."""
    g = load_golden(name)
    model, sd = build_product_model(name)  # only used as the source of state_dict keys/shapes
    assert _digest(model.state_dict()) == str(g["state_digest"][0]), "state_dict keys/shapes differ from the reference"
    cap = {}
    logits = model_forward(sd, case_input(name), CASES[name]["arch"], dtype=torch.float32, capture=cap)
    scale = np.abs(g["logits"]).max()
    assert max_abs(logits, g["logits"]) < 2e-4 * max(scale, 1.0)
    for li in (2, 3):
        ref = g[f"level{li}_out"]
        assert max_abs(cap[f"level{li}"], ref) < 2e-4 * max(np.abs(ref).max(), 1.0)
        for bi, (xb, ctb) in enumerate(cap[f"blocks{li}"]):
            ref = g[f"l{li}b{bi}_x"]
            assert max_abs(xb, ref) < 2e-4 * max(np.abs(ref).max(), 1.0), f"level {li} block {bi} x"
            if f"l{li}b{bi}_ct" in g:
                ref = g[f"l{li}b{bi}_ct"]
                assert max_abs(ctb, ref) < 2e-4 * max(np.abs(ref).max(), 1.0), f"level {li} block {bi} ct"


@pytest.mark.parametrize("name", FULL_FAST + FULL_SLOW)
def test_oracle_matches_reference_full_size(name):
    """BASELINE.json configs: oracle logits (and image-0 stage outputs) vs the reference's."""
    g = load_golden(name)
    model, sd = build_product_model(name)
    assert _digest(model.state_dict()) == str(g["state_digest"][0])
    del model
    cap = {}
    logits = model_forward(sd, case_input(name), CASES[name]["arch"], dtype=torch.float32, capture=cap)
    assert max_abs(logits, g["logits"]) < 1e-4 * max(np.abs(g["logits"]).max(), 1.0)
    for li in (2, 3):
        if f"level{li}_out" in g:
            ref = g[f"level{li}_out"]
            assert max_abs(cap[f"level{li}"][:1], ref) < 2e-4 * max(np.abs(ref).max(), 1.0)


def test_oracle_float64_agrees_with_float32():
    name = "tiny_hier"
    _, sd = build_product_model(name)
    x = case_input(name)
    a = model_forward(sd, x, CASES[name]["arch"], dtype=torch.float32)
    b = model_forward(sd, x, CASES[name]["arch"], dtype=torch.float64)
    assert max_abs(a, b) < 1e-4


def test_state_dict_keys_match_reference_for_every_entrypoint():
    """All 22 entrypoints: same state_dict keys, shapes, dtypes and parameter count as the reference."""
    import fastervit_amd
    with open(os.path.join(GOLDEN_DIR, "state_keys.json")) as f:
        ref = json.load(f)
    assert sorted(ref) == sorted(fastervit_amd.list_models())
    for name, info in ref.items():
        with torch.device("meta"):
            m = fastervit_amd.create_model(name)
        assert len(m.state_dict()) == info["keys"], name
        assert sum(p.numel() for p in m.parameters()) == info["params"], name
        assert _digest(m.state_dict()) == info["sha256"], name


def test_layout_functions_roundtrip_and_nonsquare_quirk():
    """window_partition/reverse are inverses; ct_window inverts ct_dewindow only on square grids."""
    x = torch.randn(2, 8, 6, 12)
    w = hr.window_partition(x, 3)
    assert w.shape == (2 * 2 * 4, 9, 8)
    assert torch.equal(hr.window_reverse(w, 3, 6, 12, 2), x)
    ct = torch.arange(2 * 16 * 3, dtype=torch.float32).view(2, 16, 3)
    back = hr.ct_window(hr.ct_dewindow(ct, 4, 4, 2), 4, 4, 2).reshape(2, 16, 3)
    assert torch.equal(back, ct)
    ct = torch.arange(60 * 2, dtype=torch.float32).view(1, 60, 2)
    back = hr.ct_window(hr.ct_dewindow(ct, 6, 10, 2), 6, 10, 2).reshape(1, 60, 2)
    assert not torch.equal(back, ct)
    # window 0 receives carrier ids {0, 1, 12, 13}
    assert back[0, :4, 0].div(2).tolist() == [0.0, 1.0, 12.0, 13.0]
