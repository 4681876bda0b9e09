"""Helpers shared by the test modules (test infrastructure; may import the oracle)."""
import os

import numpy as np
import torch

from tests.cases import CASES, SEED
from tests.synth import synth_input, synth_state_dict

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def build_product_model(name, device="cpu", dtype=None):
    """Product model for a parity case with the case's synthetic weights loaded."""
    import fastervit_amd
    case = CASES[name]
    model = fastervit_amd.create_model(case["entry"], **case["kwargs"]).eval()
    sd = synth_state_dict(model.state_dict(), SEED, case["family"])
    model.load_state_dict(sd, strict=True)
    if dtype is not None:
        model = model.to(dtype)
    return model.to(device), sd


def case_input(name):
    case = CASES[name]
    return synth_input(case["batch"], case["hw"][0], case["hw"][1], SEED)


def max_abs(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return (a - b).abs().max().item()


def rel_err(a, ref):
    """max |a - ref| / max |ref|"""
    ref = torch.as_tensor(ref).double()
    return max_abs(a, ref) / max(ref.abs().max().item(), 1e-30)
