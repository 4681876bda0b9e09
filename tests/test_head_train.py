"""Head-only training step (north_star training clause): oracle pinned against torch.autograd; the data-parallel reduction checked
with two gloo processes on the CPU (the four HIP kernel calls are replaced by the oracle THERE ONLY -- test infrastructure standing
in for the GPU, as in test_validate_dropin.py); the HIP kernels themselves vs torch.autograd on the GPU."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

from oracle.head_reference import head_forward_backward_ref, sgd_momentum_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _autograd_reference(feat, target, W, b, smoothing):
    W = W.clone().double().requires_grad_(True)
    b = b.clone().double().requires_grad_(True)
    logits = F.linear(feat.double(), W, b)
    loss = F.cross_entropy(logits, target, label_smoothing=smoothing)   # == timm LabelSmoothingCrossEntropy (mean of nll / smooth mix)
    loss.backward()
    return torch.cat([W.grad.reshape(-1), b.grad, loss.detach().view(1)]).float()


@pytest.mark.parametrize("smoothing", [0.0, 0.1])
def test_oracle_matches_autograd(smoothing):
    g = torch.Generator().manual_seed(0)
    B, Fdim, N = 24, 48, 37
    feat, W, b = torch.randn(B, Fdim, generator=g), torch.randn(N, Fdim, generator=g) * 0.1, torch.randn(N, generator=g) * 0.1
    tgt = torch.randint(0, N, (B,), generator=g)
    got = head_forward_backward_ref(feat, tgt, W, b, B, smoothing)
    ref = _autograd_reference(feat, tgt, W, b, smoothing)
    assert (got - ref).abs().max().item() < 1e-6
    # two shards with the global batch in the denominator sum to the full-batch gradient (what the all-reduce relies on)
    parts = head_forward_backward_ref(feat[:10], tgt[:10], W, b, B, smoothing) + head_forward_backward_ref(feat[10:], tgt[10:], W, b, B, smoothing)
    assert (parts - ref).abs().max().item() < 1e-6


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import fastervit_amd
    from fastervit_amd import dp, head_train

    # CPU stand-ins for the HIP kernel calls (test only)
    def fb(feat, target, weight, bias, grad_flat, global_batch, smoothing, scratch):
        grad_flat.copy_(head_forward_backward_ref(feat, target, weight, bias, global_batch, smoothing))
    head_train.head_forward_backward = fb
    head_train.sgd_update = sgd_momentum_ref
    d = dp.init_process_group("gloo")
    torch.manual_seed(0)
    model = fastervit_amd.create_model("faster_vit_0_224", depths=[1, 1, 1, 1], num_heads=[1, 1, 2, 4], dim=8, in_dim=8, num_classes=13)
    tr = head_train.HeadTrainer(model, lr=0.5, momentum=0.9, weight_decay=1e-3, smoothing=0.1, dist=d)
    g = torch.Generator().manual_seed(5)
    B = 12
    feat_all = torch.randn(B, tr.F, generator=g)
    tgt_all = torch.randint(0, tr.N, (B,), generator=g)
    lo, hi = dp.shard_bounds(B, world, rank)
    w0 = tr.param.clone()
    losses = [float(tr.step_on_features(feat_all[lo:hi], tgt_all[lo:hi])) for _ in range(3)]
    torch.save(dict(grad=tr.grad.clone(), param=tr.param.clone(), w0=w0, losses=losses, feat=feat_all, tgt=tgt_all, N=tr.N, F=tr.F,
                    head_w=model.head.weight.detach().clone()), os.path.join(tmp, f"r{rank}.pt"))
    d.barrier()
    d.destroy_process_group()


def test_head_gradient_allreduce_gloo_world2(tmp_path):
    """Two ranks, half the batch each, ONE all-reduce of [dW | db | loss]: the reduced gradient equals the single-process full-batch
    gradient, both ranks end with identical weights, and the weights follow torch.optim.SGD on the full batch."""
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(str(tmp_path), f"r{r}.pt")) for r in (0, 1))
    assert torch.equal(r0["param"], r1["param"]) and r0["losses"] == r1["losses"]
    N, Fd = r0["N"], r0["F"]
    # replay with torch.optim.SGD on the FULL batch in one process
    W = r0["w0"][:N * Fd].view(N, Fd).clone().requires_grad_(True)
    b = r0["w0"][N * Fd:].clone().requires_grad_(True)
    opt = torch.optim.SGD([W, b], lr=0.5, momentum=0.9, weight_decay=1e-3)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = F.cross_entropy(F.linear(r0["feat"], W, b), r0["tgt"], label_smoothing=0.1)
        loss.backward()
        last_grad = torch.cat([W.grad.reshape(-1), b.grad])
        opt.step()
        losses.append(float(loss))
    assert max(abs(a - c) for a, c in zip(losses, r0["losses"])) < 1e-5
    assert (r0["grad"][:-1] - last_grad).abs().max().item() < 1e-5           # reduced gradient of the last step == full-batch gradient
    assert (r0["param"] - torch.cat([W.detach().reshape(-1), b.detach()])).abs().max().item() < 1e-5
    assert torch.equal(r0["head_w"].reshape(-1), r0["param"][:N * Fd])        # the module's head IS the trained buffer


def test_product_head_training_has_no_cpu_path():
    import fastervit_amd
    from fastervit_amd import head_train
    torch.manual_seed(0)
    model = fastervit_amd.create_model("faster_vit_0_224", depths=[1, 1, 1, 1], num_heads=[1, 1, 2, 4], dim=8, in_dim=8, num_classes=5)
    tr = head_train.HeadTrainer(model)
    with pytest.raises(RuntimeError, match="HIP device"):
        tr.step_on_features(torch.randn(4, tr.F), torch.randint(0, 5, (4,)))


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,Fd,smoothing", [(256, 1000, 512, 0.1), (37, 21, 48, 0.0), (16, 1000, 1568, 0.1)])
def test_head_kernels_vs_autograd(B, N, Fd, smoothing):
    """fvit_head_logits / _softmax_xent / _grad through the C ABI vs torch.autograd (fp64) on the same fp32 inputs; bit-repeatable."""
    from fastervit_amd import head_train
    g = torch.Generator().manual_seed(B + N)
    feat = torch.randn(B, Fd, generator=g)
    W = torch.randn(N, Fd, generator=g) * 0.05
    b = torch.randn(N, generator=g) * 0.05
    tgt = torch.randint(0, N, (B,), generator=g)
    ref = _autograd_reference(feat, tgt, W, b, smoothing)
    grad = torch.full((N * Fd + N + 1,), float("nan"), device="cuda")
    scratch = {}
    head_train.head_forward_backward(feat.cuda(), tgt.cuda(), W.cuda(), b.cuda(), grad, B, smoothing, scratch)
    torch.cuda.synchronize()
    got = grad.cpu()
    scale = max(ref[:-1].abs().max().item(), 1e-6)
    assert (got[:-1] - ref[:-1]).abs().max().item() < 2e-5 * scale + 1e-8
    assert abs(got[-1].item() - ref[-1].item()) < 1e-5 * max(abs(ref[-1].item()), 1.0)
    grad2 = torch.empty_like(grad)
    head_train.head_forward_backward(feat.cuda(), tgt.cuda(), W.cuda(), b.cuda(), grad2, B, smoothing, scratch)
    assert torch.equal(grad, grad2)


@pytest.mark.gpu
def test_head_trainer_on_the_hip_backbone():
    """End to end on the GPU: frozen tiny backbone through the HIP HAT path, three SGD steps; loss falls, weights follow
    torch.optim.SGD on the same features, and model(x) sees the trained head (also through the 16-bit deploy plan)."""
    from fastervit_amd import head_train
    from tests.util import build_product_model, case_input
    model, _ = build_product_model("tiny_hier", "cuda")
    x = case_input("tiny_hier").cuda()
    x = torch.cat([x, x.flip(-1), x.flip(-2), x * 0.5])
    tgt = torch.tensor([1, 3, 5, 7, 9, 11, 13, 15], device="cuda")
    with torch.no_grad():
        fmax = torch.flatten(model.avgpool(model.forward_features(x)), 1).float().abs().max().item()
    LR = 0.02 / max(fmax * fmax, 1.0)      # the 'stress' fixture's pooled features are O(10): keep the step stable
    tr = head_train.HeadTrainer(model, lr=LR, momentum=0.9, weight_decay=1e-4, smoothing=0.1)
    feat = tr.features(x)
    W = tr.param[:tr.N * tr.F].view(tr.N, tr.F).detach().cpu().clone().requires_grad_(True)
    b = tr.param[tr.N * tr.F:].detach().cpu().clone().requires_grad_(True)
    opt = torch.optim.SGD([W, b], lr=LR, momentum=0.9, weight_decay=1e-4)
    losses = []
    for _ in range(3):
        losses.append(float(tr.step(x, tgt)))
        opt.zero_grad()
        F.cross_entropy(F.linear(feat.cpu(), W, b), tgt.cpu(), label_smoothing=0.1).backward()
        opt.step()
    assert losses[2] < losses[0]
    assert (tr.param.cpu() - torch.cat([W.detach().reshape(-1), b.detach()])).abs().max().item() < 2e-5 * max(W.abs().max().item(), 1.0)
    with torch.no_grad():
        logits = model(x).float()
        assert (logits.cpu() - F.linear(feat.cpu(), W, b).detach()).abs().max().item() < 1e-3 * max(logits.abs().max().item(), 1.0)
        model.switch_to_deploy(torch.float16)
        dep = model(x).float()
    assert (dep - logits).abs().max().item() < 2e-2 * max(logits.abs().max().item(), 1.0)
