"""Host-runtime behaviour on an MI355X: the bench configuration checked per stream shard
against the CPU oracle, the library-level hipGraph runner, nn.DataParallel-style replicas and concurrent host threads, device
placement errors, and a reference-shaped eval loop (validate.py:286-344) driving the HIP path.
"""
import threading

import numpy as np
import pytest
import torch

from fastervit_amd import hat_runtime
from fastervit_amd.inference import CompiledInference, PipelinedInference, accuracy_counts, evaluate
from oracle.model_reference import model_forward
from tests.cases import CASES
from tests.util import build_product_model, case_input, load_golden, max_abs

pytestmark = pytest.mark.gpu


def _native_loaded():
    with open("/proc/self/maps") as f:
        return "libfvit_hip.so" in f.read()


@pytest.mark.parametrize("streams,join_from,sizes", [(2, 3, [128, 128]), (3, None, [86, 86, 84])], ids=["timed-2shards-join3", "3shards-nojoin"])
def test_bench_configuration_every_shard_vs_oracle(streams, join_from, sizes):
    """The configuration bench.py TIMES -- deploy plan, fp16, batch 256 as 2 stream shards through levels 0-2, joined (torch.cat on the
    caller's stream behind the cross-stream join) in front of level 3, inside ONE hipGraph -- checked against the fp32 CPU oracle on 8
    images FROM EACH SHARD: logits max-abs < 1e-3; the replay is bit-repeatable and equals the eager forward of the same plan bit for bit
    (the join is where a race between the side stream's last kernel and the cat would show).  The r03 form (3 shards, no join) beside it."""
    model, sd = build_product_model("fvit0_224", "cuda")
    model = model.to(memory_format=torch.channels_last)
    g = torch.Generator(device="cpu").manual_seed(1000)
    x_cpu = torch.randn(256, 3, 224, 224, generator=g)
    x = x_cpu.cuda().contiguous(memory_format=torch.channels_last)
    runner = model.compile_inference(x, dtype=torch.float16, streams=streams, join_from=join_from)
    assert isinstance(runner, CompiledInference) and runner.graph is not None
    assert runner.plan.streams == streams and runner.plan.join_from == join_from
    y = runner(x).float().cpu().clone()
    for _ in range(4):
        assert torch.equal(y, runner(x).float().cpu())   # replays are bit-repeatable
    with torch.no_grad():
        y_eager = runner.plan.forward(x).float().cpu()    # the same plan eagerly (fork / join with events, no graph)
    assert torch.equal(y, y_eager)
    assert _native_loaded()
    assert [p.shape[0] for p in x_cpu.chunk(streams)] == sizes
    starts = [sum(sizes[:i]) for i in range(len(sizes))]
    for shard, start in enumerate(starts):
        idx = list(range(start + 3, start + 11))        # 8 images inside the shard (not only its first rows)
        ref = model_forward(sd, x_cpu[idx], CASES["fvit0_224"]["arch"])
        err = max_abs(y[idx], ref)
        print(f"bench configuration streams={streams} join_from={join_from}, shard {shard} images {idx[0]}..{idx[-1]}: logits max-abs err {err:.3e} (|logits| max {ref.abs().max():.3f})")
        assert err < 1e-3, f"shard {shard}"
    # the joined plan and a single-stream plan run the same kernels on the same rows up to the launch shape: same logits to rounding order
    y1 = model.compile_inference(x, dtype=torch.float16, streams=1, graph=False)(x).float().cpu()
    assert max_abs(y1, y) < 2e-4
    # a shorter batch through the same graph: zero-padded, sliced
    y40 = runner(x[:40]).float().cpu()
    assert y40.shape == (40, 1000) and max_abs(y40, y[:40]) < 2e-4


@pytest.mark.parametrize("depth,streams,join_from", [(2, 1, None), (3, 1, None), (2, 2, 3)], ids=["timed-2-in-flight", "3-in-flight", "2-in-flight-2shards-join3"])
def test_pipelined_steps_in_flight_vs_oracle(depth, streams, join_from):
    """The configuration bench.py TIMES since r06 -- deploy plan, fp16, batch 256 as whole-batch launches, TWO steps in flight on two streams (one hipGraph
    with its own static buffers and stage-workspace slots per runner) -- checked against the fp32 CPU oracle on 16 images of every runner's output after a
    burst of back-to-back launches; all runners bitwise equal to each other and to the single-runner graph (a shared workspace between two steps in flight
    would show here), and different inputs in flight at the same time keep their own results."""
    model, sd = build_product_model("fvit0_224", "cuda")
    model = model.to(memory_format=torch.channels_last)
    g = torch.Generator(device="cpu").manual_seed(1000)
    x_cpu = torch.randn(256, 3, 224, 224, generator=g)
    x = x_cpu.cuda().contiguous(memory_format=torch.channels_last)
    pipe = model.pipelined_inference(x, depth=depth, dtype=torch.float16, streams=streams, join_from=join_from)
    assert isinstance(pipe, PipelinedInference) and len(pipe.runners) == depth and all(r.graph is not None for r in pipe.runners)
    assert sorted(r.plan.slot_base for r in pipe.runners) == [i * streams for i in range(depth)]
    for _ in range(4 * depth + 1):          # a burst: steps overlap on the GPU
        pipe.launch()
    outs = [o.float().cpu().clone() for o in pipe.outputs()]
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    single = model.compile_inference(x, dtype=torch.float16, streams=streams, join_from=join_from)
    assert torch.equal(single(x).float().cpu(), outs[0])
    assert _native_loaded()
    idx = list(range(3, 11)) + list(range(131, 139))
    ref = model_forward(sd, x_cpu[idx], CASES["fvit0_224"]["arch"])
    err = max_abs(outs[0][idx], ref)
    print(f"pipelined depth={depth} streams={streams} join_from={join_from}: logits max-abs err {err:.3e} (|logits| max {ref.abs().max():.3f})")
    assert err < 1e-3
    # two DIFFERENT batches in flight: runner 1 gets the reversed batch while runner 0 keeps x
    pipe.wait()
    pipe.runners[1].static_x.copy_(x.flip(0))
    torch.cuda.synchronize()
    pipe._k = 0
    for _ in range(2 * depth):
        pipe.launch()
    o2 = [o.float().cpu() for o in pipe.outputs()]
    assert torch.equal(o2[0], outs[0]) and max_abs(o2[1], outs[0].flip(0)) < 2e-4   # (an image's rows sit in other tiles: same values to rounding order)


def test_compiled_inference_rejects_wrong_inputs():
    model, _ = build_product_model("tiny_hier", "cuda")
    x = case_input("tiny_hier").cuda()
    runner = model.compile_inference(x, streams=2)
    with pytest.raises(RuntimeError):
        runner(x.cpu())
    with pytest.raises(RuntimeError):
        runner(torch.cat([x, x]))
    with pytest.raises(RuntimeError):
        runner(x[..., :-1])
    with torch.no_grad():
        model.switch_to_deploy(torch.float16)
        ref = model(x).float()
    assert max_abs(runner(x).float().cpu(), ref.cpu()) < 2e-4 * max(ref.abs().max().item(), 1.0)


def test_replicas_and_threads_share_a_device_safely():
    """nn.DataParallel's mechanics on one GPU: ``replicate`` (shallow-copied __dict__, fresh broadcast parameters) + ``parallel_apply``
    (one host thread per replica).  Per-device state, a content signature for replicas and the per-stage enqueue lock make the
    replicas reproduce the original's logits; then two threads drive the ORIGINAL model concurrently on different HIP streams."""
    from torch.nn.parallel import parallel_apply, replicate
    model, _ = build_product_model("fvit0_224", "cuda")
    x = case_input("fvit0_224").cuda()
    with torch.no_grad():
        model(x)
        ref = model(x).float()
        reps = replicate(model, [0, 0])
        assert all(getattr(r, "_is_replica", False) for r in reps)
        outs = parallel_apply(reps, [(x[:4],), (x[4:],)], devices=[0, 0])
        got = torch.cat([o.float() for o in outs])
    # module mode: the conv side is MIOpen fp32 (not bit-repeatable run to run), the HAT stages are ours
    assert max_abs(got.cpu(), ref.cpu()) < 5e-4
    # DataParallel wrapper itself (single device: calls the module directly, validate.py:243-244 with num_gpu = 1)
    dp = torch.nn.DataParallel(model, device_ids=[0])
    with torch.no_grad():
        assert max_abs(dp(x).float().cpu(), ref.cpu()) < 5e-4

    # two host threads, each on its own stream, same model, same scratch slot: the runtime serialises the slot with events
    lvl = model.levels[2]
    xs = [torch.randn(6, 256, 14, 14, generator=torch.Generator().manual_seed(s)).cuda() for s in (1, 2)]
    with torch.no_grad():
        seq = [hat_runtime.stage_forward(lvl, xi).clone() for xi in xs]
    torch.cuda.synchronize()
    res, errs = [None, None], []

    def work(i):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st), torch.no_grad():
                for _ in range(4):
                    out = hat_runtime.stage_forward(lvl, xs[i])
                st.synchronize()
                res[i] = out
        except Exception as e:  # surfaced below
            errs.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    assert torch.equal(res[0], seq[0]) and torch.equal(res[1], seq[1])


def test_device_and_mode_errors():
    model, _ = build_product_model("tiny_hier", "cuda")
    x = case_input("tiny_hier").cuda()
    lvl = model.levels[2]
    C = lvl.blocks[0].attn.qkv.in_features
    xin = torch.randn(2, C, 14, 14, device="cuda")
    with pytest.raises(RuntimeError, match="no CPU"):
        lvl(xin.cpu())
    # a stage whose own parameters require grad is differentiable in eval mode with grad enabled (r05, ADVICE r04: a frozen conv side with
    # trainable HAT blocks used to get detached outputs): the gradient flows to a leaf input and to the stage's parameters
    xl = xin.clone().requires_grad_(True)
    yl = lvl(xl)
    assert yl.grad_fn is not None
    yl.sum().backward()
    assert xl.grad is not None and lvl.blocks[0].attn.qkv.weight.grad is not None
    model.zero_grad(set_to_none=True)
    # ... with the stage frozen, the forward-only entry point is what runs: a leaf input that asks for a gradient is an error, not a cut graph
    for p_ in lvl.parameters():
        p_.requires_grad_(False)
    with pytest.raises(RuntimeError, match="requires grad"):
        lvl(xin.clone().requires_grad_(True))
    for p_ in lvl.parameters():
        p_.requires_grad_(True)
    # an eval-mode whole-model forward WITHOUT torch.no_grad() runs and -- r04 -- is differentiable wherever hat_backward covers the
    # stages (tiny_hier: head_dim 24, 53-token windows): same numbers as under no_grad, and the gradient w.r.t. the input flows instead of being cut
    with torch.no_grad():
        y_ref = model(x)
    y = model(x)
    assert y.grad_fn is not None
    assert torch.equal(y.detach(), y_ref)
    xg = x.clone().requires_grad_(True)
    model(xg).sum().backward()
    assert xg.grad is not None and torch.isfinite(xg.grad).all() and xg.grad.abs().max().item() > 0
    # ... while on a geometry the kernel-sequence backward does not cover (196-token windows) a caller who asks for d/dx gets an error, not zeros,
    # and a plain grad-enabled forward still runs (one RuntimeWarning, detached stage outputs)
    import warnings
    big, _ = build_product_model("tiny_w14", "cuda")
    xb = case_input("tiny_w14").cuda()
    with pytest.raises(RuntimeError, match="requires grad"):
        big(xb.clone().requires_grad_(True))
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        yb = big(xb)
    with torch.no_grad():   # (module mode: MIOpen's fp32 convolutions are not bitwise repeatable call to call)
        assert (yb.detach() - big(xb)).abs().max().item() < 1e-3 * yb.abs().max().item()
    with pytest.raises(RuntimeError, match="no kernel-sequence backward"):
        big.enable_hat_backward(True)
    # train mode: the low-level forward-only entry point refuses (it would skip stochastic depth) ...
    model.train()
    from fastervit_amd import hat_runtime
    with pytest.raises(RuntimeError, match="inference-only"), torch.no_grad():
        hat_runtime.stage_forward(lvl, xin)
    # ... the module runs the stage with train semantics (stochastic depth switched off here: the eval numbers up to the unit kernels' rounding)
    for m in model.modules():
        if hasattr(m, "drop_prob"):
            m.drop_prob = 0.0
    with torch.no_grad():
        yt = lvl(xin)
    model.eval()
    with torch.no_grad():
        ye = lvl(xin)
    assert (yt - ye).abs().max().item() < 2e-2 * ye.abs().max().item()
    # parameters left on the CPU, input on the GPU: a clear error instead of a wild pointer
    cpu_model, _ = build_product_model("tiny_hier", "cpu")
    with pytest.raises(RuntimeError, match="parameters are on"), torch.no_grad():
        cpu_model.levels[2](xin)
    # num_classes = 0 (Identity head) under autocast takes the automatic plan and returns the pooled features
    import fastervit_amd
    torch.manual_seed(0)
    kw = dict(CASES["tiny_hier"]["kwargs"], num_classes=0)
    feat_model = fastervit_amd.create_model(CASES["tiny_hier"]["entry"], **kw).eval().cuda()
    with torch.no_grad():
        plain = feat_model(x).float()
        with torch.autocast("cuda", dtype=torch.float16):
            auto = feat_model(x).float()
    assert "_auto_plans" in feat_model.__dict__ and auto.shape == plain.shape
    assert max_abs(auto.cpu(), plain.cpu()) < 2e-2 * max(plain.abs().max().item(), 1.0)
    # a registered forward hook keeps the model on the plain nn.Module path (the plan would bypass it)
    seen = []
    h = feat_model.levels[0].register_forward_hook(lambda m, i, o: seen.append(1))
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        assert feat_model._autocast_plan(x) is None
        feat_model(x)
    h.remove()
    assert seen


def test_reference_shaped_eval_loop_on_the_hip_path():
    """validate.py:286-344 restated (fastervit_amd.inference.evaluate): synthetic loader, --amp --channels-last; the automatic deploy
    plan must engage and top-k counts must equal the CPU oracle's on the same batches; then the same loop on the captured graph."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "_shim"))
    try:
        from timm.data import create_dataset, create_loader
    finally:
        sys.path.pop(0)
    model, sd = build_product_model("fvit0_224", "cuda")
    model = model.to(memory_format=torch.channels_last)
    ds = create_dataset()
    loader = create_loader(ds, (3, 224, 224), 8, device=torch.device("cpu"))
    n, c1, c5, last = evaluate(model, loader, torch.device("cuda"), amp_dtype=torch.float16, channels_last=True)
    assert "_auto_plans" in model.__dict__ and _native_loaded()
    assert n == len(ds) and last.dtype == torch.float16
    # oracle on the same stream of batches
    rn = r1 = r5 = 0
    worst = 0.0
    logits_all = []
    for inp, tgt in loader:
        ref = model_forward(sd, inp, CASES["fvit0_224"]["arch"])
        a1, a5 = accuracy_counts(ref, tgt)
        rn, r1, r5 = rn + inp.shape[0], r1 + a1, r5 + a5
        logits_all.append(ref)
    assert (n, c1, c5) == (rn, r1, r5)
    worst = max_abs(last.float().cpu(), logits_all[-1])
    print(f"eval loop (--amp --channels-last, automatic deploy plan): last-batch logits max-abs err {worst:.3e}")
    assert worst < 1e-3
    # the same loop through the library's graph runner
    example = torch.zeros(8, 3, 224, 224, device="cuda").contiguous(memory_format=torch.channels_last)
    runner = model.compile_inference(example, streams=2)
    n2, d1, d5, last2 = evaluate(model, loader, torch.device("cuda"), channels_last=True, runner=runner)
    assert (n2, d1, d5) == (rn, r1, r5) and max_abs(last2.float().cpu(), logits_all[-1]) < 1e-3
