"""CPU-side checks of the precise deploy plan (r05; no GPU): its host packing -- every conv weight as two 16-bit terms in the [hi | lo] row layout of
fvit_conv3x3_nhwc_px, BatchNorm folded, channels zero-padded to the map layout, the K = 27 stem weights as two [64][32] terms -- and the claim the plan
rests on: on the logits a 16-bit STORED conv-side stream costs several times what the same rounding costs as a conv OPERAND."""
import torch
import torch.nn.functional as F

import fastervit_amd
from fastervit_amd.conv_runtime import DeployPlan, _fold
from oracle import model_reference as mr
from tests.cases import CASES, SEED
from tests.synth import synth_input, synth_state_dict
from tests.tools import conv_precision_sim as cs


def _model(name):
    c = CASES[name]
    m = fastervit_amd.create_model(c["entry"], **c["kwargs"]).eval()
    m.load_state_dict(synth_state_dict(m.state_dict(), SEED, c["family"]))
    return m, c


def test_dense_k_packing_drops_the_pad_channels_of_the_input():
    """r06 (fvit_conv3x3_nhwc_dense): a conv whose input map is channel-padded contracts over the real channels only: rows [Cout][terms][kd],
    column t * cv + c = w[co][tap t][c], zero tail; a conv without padding keeps the classic matrix (cv == Cin)."""
    m, _ = _model("tiny_hier")
    for terms in (1, 2):
        plan = DeployPlan(m, torch.float16)
        assert plan.dense_k
        found = 0
        for blk in m.levels[0].blocks:
            wa, _ = _fold(blk.conv1, blk.norm1)
            co, ci = wa.shape[:2]
            cop, cip = plan._cp(co), plan._cp(ci)
            wcl, wk, wband, wt, cv, wk_classic = plan._cw(wa, terms=terms)
            cv8 = (ci + 7) // 8 * 8
            if cv8 == cip:
                assert cv == cip and wk.numel() == cop * terms * 9 * cip and wk_classic is None
                continue
            found += 1
            kd = (9 * cv8 + 63) // 64 * 64
            assert cv == cv8 and wt == terms and wband is None and tuple(wk.shape) == (cop, terms * kd)
            assert tuple(wk_classic.shape) == (cop, terms * 9 * cip)    # the classic rows ride along for the patch form of the kernel
            hi = wk[:, :9 * cv].float().view(cop, 3, 3, cv)
            assert wk[:, 9 * cv:kd].abs().sum().item() == 0 and hi[co:].abs().sum().item() == 0 and hi[..., ci:].abs().sum().item() == 0
            ref = wa.to(torch.float16).float().permute(0, 2, 3, 1)
            assert torch.equal(hi[:co, :, :, :ci], ref)
            if terms == 2:
                lo = wk[:, kd:kd + 9 * cv].float().view(cop, 3, 3, cv)
                assert ((hi + lo)[:co, :, :, :ci] - wa.permute(0, 2, 3, 1)).abs().max().item() <= 2.0 ** -20 * wa.abs().max().item()
                assert wk[:, kd + 9 * cv:].abs().sum().item() == 0
        if plan._cp(m.levels[0].blocks[0].conv1.weight.shape[1]) != (m.levels[0].blocks[0].conv1.weight.shape[1] + 7) // 8 * 8:
            assert found > 0
    sig = plan._signature()
    plan.dense_k = False
    assert plan._signature() != sig


def test_precise_plan_packs_every_conv_weight_as_two_terms():
    m, _ = _model("tiny_hier")
    plan = DeployPlan(m, torch.float16)
    plan.precise = True
    plan.dense_k = False   # the classic [Cout][3][3][Cin padded] rows (the dense-K packing: test_dense_k_packing_drops_the_pad_channels_of_the_input)
    plan._build()
    t = plan.t
    lvl0 = m.levels[0].blocks[0]
    wa, ba = _fold(lvl0.conv1, lvl0.norm1)                       # fp32 folded conv1 + BN of the first ConvBlock
    (wcl, wk, wband, terms, cv, _classic), bias, _, _ = t["levels"][0]["blocks"][0]
    co, ci = wa.shape[:2]
    cop, cip = plan._cp(co), plan._cp(ci)
    assert terms == 2 and wband is None and cv == cip and tuple(wk.shape) == (cop, 2 * 9 * cip) and wk.dtype == torch.float16
    hi = wk[:, :9 * cip].float().view(cop, 3, 3, cip)
    lo = wk[:, 9 * cip:].float().view(cop, 3, 3, cip)
    rec = (hi + lo)[:co, :, :, :ci].permute(0, 3, 1, 2)
    assert (rec - wa).abs().max().item() <= 2.0 ** -20 * wa.abs().max().item()          # two fp16 terms: ~22 bits
    assert (hi[:co, :, :, :ci].permute(0, 3, 1, 2) - wa).abs().max().item() > 50 * (rec - wa).abs().max().item()   # ... against 11 for one
    assert hi[co:].abs().max().item() == 0 and lo[:, :, :, ci:].abs().max().item() == 0  # pad channels stay exactly zero
    assert torch.equal(bias[:co], ba) and bias[co:].abs().max().item() == 0
    # the three Downsample convs and the second stem conv too
    for e in t["levels"]:
        if "down" in e:
            assert e["down"][3][3] == 2
    assert t["stem"][2][3] == 2
    # standard stem (3 -> 64): the K = 27 weights as two [64][32] terms
    m0, _ = _model("fvit0_224")
    p0 = DeployPlan(m0, torch.float16)
    p0.precise = True
    p0._build()
    w0, _ = _fold(m0.patch_embed.conv_down[0], m0.patch_embed.conv_down[1])
    k = (p0.t["stem_k"].float() + p0.t["stem_k_lo"].float())[:, :27].view(64, 3, 3, 3).permute(0, 3, 1, 2)
    assert (k - w0).abs().max().item() <= 2.0 ** -20 * w0.abs().max().item()
    assert p0.t["stem_k"][:, 27:].abs().max().item() == 0
    # the options are part of the plan's signature: flipping `precise` re-folds
    sig = p0._signature()
    p0.precise = False
    assert p0._signature() != sig


def test_stored_streams_cost_more_than_rounded_operands():
    """tests/tools/conv_precision_sim.py on a small model (4 images): the plan's premise, pinned on CPU."""
    m, c = _model("tiny_hier")
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = synth_input(4, *c["hw"], seed=SEED)
    exact = {k: "f" for k in cs.BASE}
    ref = cs.forward(sd, x, c["arch"], exact)
    assert torch.equal(ref, mr.model_forward(sd, x, c["arch"]))          # the replay with nothing rounded IS the oracle

    def err(**over):
        return (cs.forward(sd, x, c["arch"], {**exact, **over}) - ref).abs().max().item()

    stored = err(res="r") + err(down="r") + err(hat="r")      # the streams the 16-bit plan keeps in fp16
    operand = err(mid="r")                                     # a pure operand rounding (the ConvBlock's inner activation)
    plan16 = err(**{k: "r" for k in ("img", "stem", "mid", "res", "ln", "down", "hat", "wstem")}, w="d")
    precise = err(mid="r", stem0="r")                          # what the precise plan still rounds once: operands only
    print(f"stored streams {stored:.2e}, one operand {operand:.2e}, 16-bit plan {plan16:.2e}, precise plan {precise:.2e}")
    assert stored > 3 * operand and precise < 0.5 * plan16   # tiny_hier (stress weights): 1.0e-3 vs 1.7e-4; 2.8e-4 vs 9.5e-4 (FasterViT-4: 1.3e-4 vs 6.7e-4)
