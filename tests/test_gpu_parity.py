"""End-to-end parity on an MI355X: HIP HAT path vs the reference's golden vectors and the CPU oracle.

Tolerances (stated per north_star "logits max-abs < 1e-3"; r02: tightened to the measured margins):
  * fp16 MFMA operands, fp32 accumulate / residual / LayerNorm / softmax (the default): logits max-abs error < 1e-3 for FasterViT-0
    (|logits| <= 1.2) in module mode (conv side fp32), in deploy mode (own fp16 conv kernels), under autocast (automatic plan) and with
    stream shards; relative error < 1.5e-3 for the variants whose logits reach 5-9 (FasterViT-4, any-res, tiny fixtures); stage maps and
    the 'stress' fixtures relative < 1e-3 (stages); per block < 1.2e-3 (x) / 1.5e-3 (carrier tokens), tiny-model logits < 1.6e-3, stress logits < 1.1e-3:
    r06, ~2x the measured errors (profiles/r06_parity_margins_measured.log).
  * bf16 operands: 8 mantissa bits per operand, measured 3.3e-3 on FasterViT-0: asserted < 5e-3, does NOT meet the bar.
"""
import numpy as np
import pytest
import torch

from oracle import hat_reference as hr
from oracle.model_reference import model_forward
from tests.cases import CASES
from tests.util import build_product_model, case_input, load_golden, max_abs, rel_err

pytestmark = pytest.mark.gpu

TINY = [n for n, c in CASES.items() if c["per_block"]]


def _native_loaded():
    with open("/proc/self/maps") as f:
        return "libfvit_hip.so" in f.read()


@pytest.mark.parametrize("name", TINY)
def test_tiny_models_vs_reference_goldens(name):
    """Small configs ('stress' weights): stage outputs and logits vs the reference."""
    g = load_golden(name)
    model, _ = build_product_model(name, "cuda")
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
    assert _native_loaded()
    # r06: bounds at ~2x the measurement (profiles/r06_parity_margins_measured.log: level outputs <= 9.7e-4, logits <= 8.0e-4 over the eight fixtures; r05: 5e-3)
    for li in (2, 3):
        assert rel_err(feats[li], g[f"level{li}_out"]) < 2e-3, f"level {li} output"
    assert rel_err(logits, g["logits"]) < 1.6e-3


@pytest.mark.parametrize("name", TINY)
def test_hat_blocks_vs_reference_goldens(name):
    """HAT.forward(x, ct) block by block (fvit_hat_block_forward), fed with the reference's own block inputs."""
    g = load_golden(name)
    model, _ = build_product_model(name, "cuda")
    case = CASES[name]
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
            ref_x = g[f"l{li}b{bi}_x"]
            # r06: ~2x the measured per-block error (x <= 5.7e-4, carrier tokens <= 7.0e-4 over the eight fixtures; r05 asserted 5e-3)
            assert rel_err(xo.cpu(), ref_x) < 1.2e-3, f"{name} level {li} block {bi} x"
            if ct is not None:
                ref_ct = g[f"l{li}b{bi}_ct"]
                assert rel_err(cto.cpu(), ref_ct) < 1.5e-3, f"{name} level {li} block {bi} ct"
                ct = torch.from_numpy(ref_ct)
            xw = torch.from_numpy(ref_x)  # next block starts from the reference's state: errors do not compound


def test_fvit0_224_logits_vs_reference_fp16_operands():
    """BASELINE config: faster_vit_0_224, batch 8, 'init' weights: logits max-abs < 1e-3 (conv side fp32)."""
    g = load_golden("fvit0_224")
    model, _ = build_product_model("fvit0_224", "cuda")
    x = case_input("fvit0_224").cuda()
    with torch.no_grad():
        logits = model(x).float().cpu()
    err = max_abs(logits, g["logits"])
    print(f"faster_vit_0_224 fp16-operand logits max-abs err {err:.3e} (|logits| max {np.abs(g['logits']).max():.3f})")
    assert err < 1e-3


def test_fvit0_224_channels_last_and_autocast():
    """Same model with the conv side in channels_last + fp16 autocast (the bench configuration)."""
    g = load_golden("fvit0_224")
    model, _ = build_product_model("fvit0_224", "cuda")
    model = model.to(memory_format=torch.channels_last)
    x = case_input("fvit0_224").cuda().contiguous(memory_format=torch.channels_last)
    assert model.auto_deploy
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        out = model(x)
        assert out.dtype == torch.float16 and "_auto_plans" in model.__dict__   # validate.py --amp: the automatic deploy plan
        logits = out.float().cpu()
        model.auto_deploy = False
        plain = model(x).float().cpu()                                           # plain nn.Module path (MIOpen convs under autocast)
    err, err_plain = max_abs(logits, g["logits"]), max_abs(plain, g["logits"])
    print(f"faster_vit_0_224 autocast-fp16 + channels_last logits max-abs err {err:.3e} (auto deploy plan), {err_plain:.3e} (module path)")
    # the automatic plan (own fp16 conv kernels) meets the north-star bar; the plain module path's fp16 convolutions are MIOpen's (1.1e-3)
    assert err < 1e-3 and err_plain < 2e-3
    # outside autocast, with grad enabled, or in train mode nothing is switched automatically
    model.auto_deploy = True
    assert model._autocast_plan(x) is None
    with torch.autocast("cuda", dtype=torch.float16):
        assert model._autocast_plan(x) is None                                  # grad enabled
        with torch.no_grad():
            assert model._autocast_plan(x) is not None


def test_fvit0_224_stage_maps_vs_reference():
    """HAT stages alone on the reference's own stage inputs (image 0 of the golden batch)."""
    for case in ("fvit0_224", "fvit0_224_stress"):
        g = load_golden(case)
        model, _ = build_product_model(case, "cuda")
        for li in (2, 3):
            lvl = model.levels[li]
            ds, lvl.downsample = lvl.downsample, None
            with torch.no_grad():
                out = lvl(torch.from_numpy(g[f"level{li}_in"]).cuda()).cpu()
            lvl.downsample = ds
            e = rel_err(out, g[f"level{li}_out"])
            print(f"{case} level {li} stage rel err {e:.3e}")
            assert e < 1e-3   # measured 2.3e-4 ... 4.9e-4


def test_fvit0_224_stress_logits():
    g = load_golden("fvit0_224_stress")
    model, _ = build_product_model("fvit0_224_stress", "cuda")
    with torch.no_grad():
        logits = model(case_input("fvit0_224_stress").cuda()).float().cpu()
    e = rel_err(logits, g["logits"])
    print(f"faster_vit_0_224 stress-weights logits rel err {e:.3e}")
    assert e < 1.1e-3   # r06: 2x the measured 5.3e-4 (r05: 5e-3)


def test_fvit0_224_bf16_operands():
    g = load_golden("fvit0_224")
    model, _ = build_product_model("fvit0_224", "cuda")
    model.set_hat_operand_dtype("bf16")
    with torch.no_grad():
        logits = model(case_input("fvit0_224").cuda()).float().cpu()
    err = max_abs(logits, g["logits"])
    print(f"faster_vit_0_224 bf16-operand logits max-abs err {err:.3e}")
    assert err < 5e-3   # 8 mantissa bits per operand: does not meet the 1e-3 bar; measured 3.3e-3


def test_fvit4_224_logits_vs_reference():
    """faster_vit_4_224 (head_dim 49 -> padded 64, layer scale, propagation), batch 2."""
    g = load_golden("fvit4_224")
    model, _ = build_product_model("fvit4_224", "cuda")
    with torch.no_grad():
        logits = model(case_input("fvit4_224").cuda()).float().cpu()
    err = max_abs(logits, g["logits"])
    print(f"faster_vit_4_224 logits max-abs err {err:.3e} (|logits| max {np.abs(g['logits']).max():.3f})")
    # the FAST mode (16-bit operands rounded once) is claimed RELATIVE on this model (|logits| 7): ~5e-3 absolute, above north_star's absolute 1e-3 --
    # which the x3 modes meet with margin (asserted absolute in the second half of this test and in tests/test_gpu_x3.py)
    assert err < 1e-3 * max(np.abs(g["logits"]).max(), 1.0)
    model.set_hat_operand_dtype("f16x3")
    with torch.no_grad():
        err3 = max_abs(model(case_input("fvit4_224").cuda()).float().cpu(), g["logits"])
    print(f"faster_vit_4_224 f16x3 logits max-abs err {err3:.3e} ABSOLUTE")
    assert err3 < 2e-4   # north_star: < 1e-3; measured 2.6e-5


@pytest.mark.parametrize("name", ["fvit4_21k_384", "fvit4_21k_768"])
def test_fvit4_21k_long_window_logits_vs_reference(name):
    """faster_vit_4_21k_{384,768}: stage 2 is ONE window of 576 / 2304 tokens per image (stage 3: 144 / 576) -> the online-softmax
    attention kernel with the compact relative-bias table; logits vs the reference CPU forward on the same weights."""
    g = load_golden(name)
    model, _ = build_product_model(name, "cuda")
    with torch.no_grad():
        logits = model(case_input(name).cuda()).float().cpu()
    assert _native_loaded()
    err = max_abs(logits, g["logits"])
    print(f"{CASES[name]['entry']} logits max-abs err {err:.3e} (|logits| max {np.abs(g['logits']).max():.3f})")
    assert err < 1e-3 * max(np.abs(g["logits"]).max(), 1.0)


@pytest.mark.parametrize("entry,kwargs,hw", [
    ("faster_vit_1_224", {}, (224, 224)), ("faster_vit_2_224", {}, (224, 224)), ("faster_vit_3_224", {}, (224, 224)),
    ("faster_vit_5_224", {}, (224, 224)), ("faster_vit_6_224", {}, (224, 224)),        # head_dim 80 -> 96-wide head padding
    ("faster_vit_4_21k_224", {}, (224, 224)), ("faster_vit_4_21k_512", {}, (512, 512)),  # stage 2: 1024 tokens, stage 3: 256
    ("faster_vit_0_any_res", dict(resolution=[200, 312]), (200, 312)),
    ("faster_vit_4_21k_384_any_res", dict(resolution=[384, 384]), (384, 384))])
def test_remaining_entrypoints_run_on_the_hip_path(entry, kwargs, hw):
    """Every variant family of the reference's registry runs through the HIP HAT path (random-init weights, batch 1): finite logits,
    repeatable, and equal to the CPU oracle on the same weights for the variants small enough to run it here in seconds."""
    import fastervit_amd
    torch.manual_seed(0)
    model = fastervit_amd.create_model(entry, **kwargs).eval()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.cuda()
    x = torch.randn(1, 3, *hw, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        model(x.cuda())  # MIOpen's first call of a conv configuration may pick a different algorithm than the later ones
        a = model(x.cuda()).float().cpu()
        b = model(x.cuda()).float().cpu()
    assert _native_loaded()
    assert a.shape == (1, 1000) and torch.isfinite(a).all()
    # MIOpen's fp32 convolutions (module mode) are not bitwise repeatable run to run (observed: 1.0e-4 on |logits| = 1.0)
    assert max_abs(a, b) <= 5e-4 * max(a.abs().max().item(), 1.0)
    if entry in ("faster_vit_1_224", "faster_vit_2_224", "faster_vit_0_any_res"):
        from fastervit_amd.models.faster_vit import _ARCH
        v = entry.split("_")[2]
        arch = dict(depths=_ARCH[v]["depths"], num_heads=_ARCH[v]["num_heads"], window_size=_ARCH[v]["window_size"], ct_size=2,
                    dim=_ARCH[v]["dim"], resolution=kwargs.get("resolution", 224), hat=_ARCH[v]["hat"], do_propagation=_ARCH[v]["ls"],
                    layer_norm_last=False, any_res=entry.endswith("any_res"))
        ref = model_forward(sd, x, arch)
        err = max_abs(a, ref)
        print(f"{entry} logits max-abs err vs CPU oracle {err:.3e} (|logits| max {ref.abs().max().item():.3f})")
        assert err < 2e-3 * max(ref.abs().max().item(), 1.0)


def test_fvit4_anyres_576x960_logits_vs_reference():
    """faster_vit_4_any_res 576x960, ws [7,7,12,6], ct 2: non-square carrier grid (G = 60, S = 148)."""
    g = load_golden("fvit4_anyres_576x960")
    model, _ = build_product_model("fvit4_anyres_576x960", "cuda")
    with torch.no_grad():
        logits = model(case_input("fvit4_anyres_576x960").cuda()).float().cpu()
    err = max_abs(logits, g["logits"])
    print(f"faster_vit_4_any_res 576x960 logits max-abs err {err:.3e} (|logits| max {np.abs(g['logits']).max():.3f})")
    assert err < 1e-3 * max(np.abs(g["logits"]).max(), 1.0)   # fast mode: relative claim (see test_fvit4_224_logits_vs_reference)
    model.set_hat_operand_dtype("f16x3")
    with torch.no_grad():
        err3 = max_abs(model(case_input("fvit4_anyres_576x960").cuda()).float().cpu(), g["logits"])
    print(f"faster_vit_4_any_res 576x960 f16x3 logits max-abs err {err3:.3e} ABSOLUTE")
    assert err3 < 2e-4   # north_star: < 1e-3; measured 2.4e-5


def test_batch_256_properties():
    """BASELINE batch size (256): size-independent properties instead of a CPU oracle run --
    (1) images are independent: the first 8 of 256 reproduce the batch-8 logits bit for bit up to fp16 noise,
    (2) a permutation of the batch permutes the logits, (3) repeat calls are deterministic."""
    g = load_golden("fvit0_224")
    model, _ = build_product_model("fvit0_224", "cuda")
    x8 = case_input("fvit0_224").cuda()
    gen = torch.Generator(device="cpu").manual_seed(3)
    x = torch.cat([x8, torch.randn(248, 3, 224, 224, generator=gen).cuda()])
    with torch.no_grad():
        y = model(x).float()
        y2 = model(x).float()
        perm = torch.randperm(256, generator=gen).cuda()
        yp = model(x[perm]).float()
    assert torch.equal(y, y2)
    assert max_abs(y[:8].cpu(), g["logits"]) < 1e-3
    assert max_abs(yp.cpu(), y[perm].cpu()) < 2e-4
    assert torch.isfinite(y).all()


def test_oracle_on_gpu_box_matches_goldens():
    """The oracle itself, executed on the GPU box's host CPU, still reproduces the reference goldens."""
    name = "tiny_anyres"
    g = load_golden(name)
    _, sd = build_product_model(name)
    logits = model_forward(sd, case_input(name), CASES[name]["arch"])
    assert max_abs(logits, g["logits"]) < 2e-4 * np.abs(g["logits"]).max()


@pytest.mark.parametrize("name,tol", [("fvit0_224", 1e-3), ("tiny_hier", None), ("tiny_anyres", None), ("tiny_w14", None),
                                      ("tiny_21k_384", None), ("tiny_anyres_w16", None), ("tiny_d80", None), ("fvit4_224", None)])
def test_deploy_mode_vs_reference(name, tol):
    """switch_to_deploy(): BN folded into the convs, fp16 channels_last conv side, fused glue kernels."""
    g = load_golden(name)
    model, _ = build_product_model(name, "cuda")
    model.switch_to_deploy(torch.float16)
    with torch.no_grad():
        logits = model(case_input(name).cuda()).float().cpu()
    err, rel = max_abs(logits, g["logits"]), rel_err(logits, g["logits"])
    print(f"{name} deploy-mode (fp16 conv side) logits max-abs err {err:.3e}, relative {rel:.3e}")
    # absolute 1e-3 for FasterViT-0 (|logits| <= 1.2); relative 1.5e-3 for the variants whose logits reach 5-9 (measured 5.1e-4 ... 8.2e-4)
    assert (err < tol) if tol is not None else (rel < 1.5e-3)
    # a weight update is picked up (plan is rebuilt from the new parameter versions)
    with torch.no_grad():
        model.head.bias.add_(1.0)
        logits2 = model(case_input(name).cuda()).float().cpu()
    assert max_abs(logits2 - 1.0, logits) < 1e-3
    model.switch_to_deploy(None)
    with torch.no_grad():
        logits3 = model(case_input(name).cuda()).float().cpu()
    assert rel_err(logits3 - 1.0, g["logits"]) < 5e-3


@pytest.mark.parametrize("streams", [2, 3])
def test_deploy_mode_stream_shards(streams):
    """Deploy mode with the batch run as shards on several HIP streams: same logits as the single-stream plan
    (images are independent), inside and outside a hipGraph."""
    g = load_golden("fvit0_224")
    model, _ = build_product_model("fvit0_224", "cuda")
    x = case_input("fvit0_224").cuda()
    model.switch_to_deploy(torch.float16)
    with torch.no_grad():
        ref = model(x).float()
    model.switch_to_deploy(torch.float16, streams=streams)
    with torch.no_grad():
        y = model(x).float()
        y2 = model(x).float()
    assert max_abs(y.cpu(), ref.cpu()) < 2e-4 and torch.equal(y, y2)
    assert max_abs(y.cpu(), g["logits"]) < 1e-3
    # capturable: fork / join of the side streams inside one graph
    static_x = x.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        model(static_x)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph), torch.no_grad():
        static_y = model(static_x)
    graph.replay()
    torch.cuda.synchronize()
    assert max_abs(static_y.float().cpu(), y.cpu()) < 2e-4


def test_deploy_plan_options_agree():
    """The deploy plan's switches change the kernels, not the result: fused two-conv stem vs stem + stride-2 conv (bit-identical),
    channel-padded maps vs the MIOpen fallback for 24-channel maps, unequal shard sizes, the free-running ShardRunner, and
    stage_forward writing into a strided view of a padded map."""
    from fastervit_amd import hat_runtime
    from fastervit_amd.conv_runtime import ShardRunner
    model, _ = build_product_model("fvit0_224", "cuda")
    x = case_input("fvit0_224").cuda()
    model.switch_to_deploy(torch.float16)
    plan = model.__dict__["_deploy_plan"]
    with torch.no_grad():
        ref = model(x).float()
        plan.fused_stem = False
        two = model(x).float()
        plan.fused_stem = True
        assert torch.equal(ref, two)
        plan.streams, plan.shard_sizes = 3, [3, 1, 4]
        uneven = model(x).float()
        plan.streams, plan.shard_sizes = 1, None
        assert max_abs(uneven.cpu(), ref.cpu()) < 2e-4
        runner = ShardRunner(plan, x, 2)
        runner.launch()
        runner.launch()
        assert max_abs(runner.outputs().float().cpu(), ref.cpu()) < 2e-4
    # 24 / 48-channel conv side: padded to 64 (HIP conv kernels) vs unpadded (MIOpen + glue kernels)
    g = load_golden("tiny_hier")
    small, _ = build_product_model("tiny_hier", "cuda")
    xs = case_input("tiny_hier").cuda()
    small.switch_to_deploy(torch.float16)
    with torch.no_grad():
        a = small(xs).float().cpu()
        small.switch_to_deploy(torch.float16)
        small.__dict__["_deploy_plan"].pad_channels = False
        b = small(xs).float().cpu()
    assert rel_err(a, g["logits"]) < 1e-2 and rel_err(b, g["logits"]) < 1e-2 and rel_err(a, b) < 5e-3
    # stage_forward with strided input / output views (first C channels of wider maps)
    lvl = small.levels[2]
    C = lvl.blocks[0].attn.qkv.in_features
    xin = torch.randn(2, C + 40, 14, 14, device="cuda", dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    out = torch.zeros(2, C + 24, 14, 14, device="cuda", dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        dense = hat_runtime.stage_forward(lvl, xin[:, :C].contiguous(memory_format=torch.channels_last))
        hat_runtime.stage_forward(lvl, xin[:, :C], out=out[:, :C])
    assert torch.equal(out[:, :C], dense) and out[:, C:].abs().max().item() == 0.0
