"""Pins the claim DESIGN.md section 2 rests on, on CPU: in the logits of a FasterViT the rounding of the WEIGHTS dominates the rounding of
the activations, so the second 16-bit term of the x2 operand modes goes to the weights (tests/tools/precision_sim.py replays the fp32
oracle with the matmul operands rounded the way the HIP kernels round them)."""
import torch

import fastervit_amd
from oracle import hat_reference as hr
from oracle import model_reference as mr
from tests.cases import CASES, SEED
from tests.synth import synth_input, synth_state_dict
from tests.tools import precision_sim as ps


def test_weight_rounding_dominates_activation_rounding():
    c = CASES["tiny_hier"]
    model = fastervit_amd.create_model(c["entry"], **c["kwargs"])
    sd = synth_state_dict(model.state_dict(), SEED, "init")
    x = synth_input(4, *c["hw"], seed=SEED)
    keep = (hr.window_attention, hr.mlp)
    try:
        hr.window_attention, hr.mlp = ps.window_attention, ps.mlp
        B = torch.bfloat16
        ps.MODE = ps.Mode()
        ref = mr.model_forward(sd, x, c["arch"])
        errs = {}
        for name, mode in (("bf16", ps.Mode(B)), ("split_a", ps.Mode(B, 2, 1)), ("split_w", ps.Mode(B, 1, 2)), ("both", ps.Mode(B, 2, 2))):
            ps.MODE = mode
            errs[name] = (mr.model_forward(sd, x, c["arch"]) - ref).abs().max().item()
    finally:
        hr.window_attention, hr.mlp = keep
        ps.MODE = ps.Mode()
    print(errs)
    assert errs["split_w"] < 0.5 * errs["bf16"]          # a second weight term removes most of the error ...
    assert errs["split_a"] > 0.7 * errs["bf16"]          # ... a second activation term next to none of it
    assert errs["both"] < 0.1 * errs["bf16"]
