"""Head-only training step on the frozen MI355X HAT backbone (north_star's training clause; SURVEY.md §8f-4, narrowed).

What the reference does (fastervit/train.py): wraps the whole model in DistributedDataParallel (train.py:542-551), so every backward
all-reduces all 31.4 M gradients, and reduces the loss across ranks for logging (train.py:910).  What is built here is the slice
north_star names: "a single RCCL all-reduce over xGMI on the classifier gradient/loss only" --

  features   x -> pooled, normalised features of the FROZEN backbone through the HIP forward path (eval mode: BatchNorm running
             statistics, DropPath / Dropout identity); no backward through the HAT stages exists
  forward    logits = feat . W^T + b                                  (FasterViT.head, FV:927, 959)   fvit_head_logits
  loss       label-smoothed cross entropy (train.py:685 / 687)                                        fvit_head_softmax_xent
  backward   [dW | db | loss] in ONE flat fp32 buffer, scaled by 1 / global_batch                     fvit_head_grad
  reduce     ONE all-reduce (SUM) of that buffer: 513 001 floats = 2.05 MB for FasterViT-0 (RCCL over xGMI on GPUs, gloo in the
             CPU tests): afterwards every rank holds the global-batch gradient and the global mean loss
  update     SGD with momentum / weight decay on the flat [W | b] parameter buffer             fvit_sgd_momentum

The arithmetic of forward / loss / backward / update is in libfvit_hip.so (csrc/fvit_head.hip, exact-fp32 MFMA, no atomics); this
module owns buffers and sequencing only.  There is no CPU path: CPU tensors raise (the gloo test substitutes the oracle for the four
kernel calls, as test infrastructure).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def head_forward_backward(feat: torch.Tensor, target: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, grad_flat: torch.Tensor,
                          global_batch: int, smoothing: float, scratch: dict) -> None:
    """Kernel sequence of one local step: fills ``grad_flat`` = [dW | db | loss] (already divided by ``global_batch``)."""
    if not feat.is_cuda:
        raise RuntimeError("head training runs only on a HIP device (libfvit_hip.so kernels); there is no CPU fallback")
    B, F = feat.shape
    N = weight.shape[0]
    for t, name in ((feat, "feat"), (weight, "weight"), (bias, "bias"), (grad_flat, "grad_flat")):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.device != feat.device:
            raise RuntimeError(f"head_forward_backward: {name} must be a contiguous fp32 tensor on {feat.device}")
    if target.dtype != torch.int64 or target.device != feat.device or target.numel() != B:
        raise RuntimeError("head_forward_backward: target must be an int64 tensor of B class indices on the feature device")
    if grad_flat.numel() != N * F + N + 1:
        raise RuntimeError(f"grad_flat must hold N*F + N + 1 = {N * F + N + 1} floats")
    key = (B, N, str(feat.device))
    if scratch.get("key") != key:
        scratch["key"] = key
        scratch["logits"] = torch.empty(B, N, dtype=torch.float32, device=feat.device)
        scratch["loss_rows"] = torch.empty(B, dtype=torch.float32, device=feat.device)
        scratch["row_stats"] = torch.empty(2 * B, dtype=torch.float32, device=feat.device)
    lib = _lib.lib()
    logits, loss_rows, row_stats = scratch["logits"], scratch["loss_rows"], scratch["row_stats"]
    with torch.cuda.device(feat.device):
        st = torch.cuda.current_stream(feat.device).cuda_stream
        inv = C.c_float(1.0 / float(global_batch))
        _lib.check(lib.fvit_head_logits(feat.data_ptr(), weight.data_ptr(), _ptr(bias), logits.data_ptr(), B, N, F, st), "fvit_head_logits")
        _lib.check(lib.fvit_head_softmax_xent(logits.data_ptr(), target.data_ptr(), loss_rows.data_ptr(), row_stats.data_ptr(), B, N,
                                              C.c_float(smoothing), inv, st), "fvit_head_softmax_xent")
        _lib.check(lib.fvit_head_grad(logits.data_ptr(), feat.data_ptr(), loss_rows.data_ptr(), grad_flat.data_ptr(), B, N, F, inv, st),
                   "fvit_head_grad")


def sgd_update(param_flat: torch.Tensor, momentum_flat: torch.Tensor, grad_flat: torch.Tensor, n: int, lr: float, mu: float, wd: float) -> None:
    if not param_flat.is_cuda:
        raise RuntimeError("head training runs only on a HIP device; there is no CPU fallback")
    with torch.cuda.device(param_flat.device):
        st = torch.cuda.current_stream(param_flat.device).cuda_stream
        _lib.check(_lib.lib().fvit_sgd_momentum(param_flat.data_ptr(), momentum_flat.data_ptr(), grad_flat.data_ptr(), n, C.c_float(lr),
                                                C.c_float(mu), C.c_float(wd), st), "fvit_sgd_momentum")


def _bump_version(t: torch.Tensor) -> None:
    inc = getattr(torch.autograd.graph, "increment_version", None)
    if inc is not None:
        inc(t)
    else:  # older torch: an in-place no-op bumps the shared version counter of the buffer and its views
        t.add_(0)


class HeadTrainer:
    """Data-parallel fine-tuning of ``model.head`` on the frozen backbone.

    ``dist``: the ``torch.distributed`` module of an initialised process group (``fastervit_amd.dp.init_process_group``) or None
    for a single process.  Every rank must construct the trainer from identical head weights (same seed / checkpoint); the
    all-reduced gradient then keeps them identical -- the invariant DDP maintains by construction (train.py:551).
    """

    def __init__(self, model, lr: float = 0.1, momentum: float = 0.9, weight_decay: float = 0.0, smoothing: float = 0.1, dist=None,
                 amp_dtype: Optional[torch.dtype] = None):
        if not isinstance(model.head, torch.nn.Linear):
            raise RuntimeError("HeadTrainer needs a Linear classifier head (num_classes > 0)")
        self.model = model.eval()       # frozen backbone: eval semantics (BN running stats, no DropPath); also what the HIP path requires
        for p in model.parameters():
            p.requires_grad_(False)
        self.lr, self.mu, self.wd, self.smoothing = float(lr), float(momentum), float(weight_decay), float(smoothing)
        self.dist = dist
        self.world = dist.get_world_size() if dist is not None else 1
        self.amp_dtype = amp_dtype
        head = model.head
        self.N, self.F = head.weight.shape
        dev = head.weight.device
        # flat [W | b] buffer; the module's parameters become views of it, so the update is visible to model(x) / state_dict()
        self.param = torch.empty(self.N * self.F + self.N, dtype=torch.float32, device=dev)
        self.param[:self.N * self.F].copy_(head.weight.detach().float().reshape(-1))
        self.param[self.N * self.F:].copy_(head.bias.detach().float() if head.bias is not None else torch.zeros(self.N, device=dev))
        head.weight.data = self.param[:self.N * self.F].view(self.N, self.F)
        if head.bias is None:
            head.bias = torch.nn.Parameter(torch.zeros(self.N, device=dev), requires_grad=False)
        head.bias.data = self.param[self.N * self.F:]
        self.mom = torch.zeros_like(self.param)
        self.grad = torch.zeros(self.N * self.F + self.N + 1, dtype=torch.float32, device=dev)   # [dW | db | loss]
        self._scratch = {}
        self.steps = 0

    # ---- frozen backbone -----------------------------------------------------------------
    @torch.no_grad()
    def features(self, x: torch.Tensor) -> torch.Tensor:
        """(B, F) fp32: avgpool(norm(forward_features)) -- FV:953-958 without the head.  Runs the HIP forward path (module mode, or the
        automatic 16-bit plan when ``amp_dtype`` is set)."""
        m = self.model
        if self.amp_dtype is not None and x.is_cuda:
            with torch.autocast("cuda", dtype=self.amp_dtype):
                f = m.forward_features(x)
        else:
            f = m.forward_features(x)
        return torch.flatten(m.avgpool(f), 1).float().contiguous()

    # ---- one optimisation step -----------------------------------------------------------
    def step(self, x: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        """One data-parallel SGD step on this rank's shard ``(x, target)``; returns the global mean loss (0-dim tensor)."""
        return self.step_on_features(self.features(x), target)

    def step_on_features(self, feat: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        B = feat.shape[0]
        global_batch = B * self.world          # equal shards (the sampler's contract, as under DDP)
        W = self.param[:self.N * self.F].view(self.N, self.F)
        b = self.param[self.N * self.F:]
        head_forward_backward(feat, target, W, b, self.grad, global_batch, self.smoothing, self._scratch)
        if self.dist is not None:
            self.dist.all_reduce(self.grad, op=self.dist.ReduceOp.SUM)     # THE collective of the training path: gradient + loss together
        sgd_update(self.param, self.mom, self.grad, self.N * self.F + self.N, self.lr, self.mu, self.wd)
        # the kernel wrote through a raw pointer: tell torch (and the deploy plan's weight signature, which keys on _version) so
        for t in (self.param, self.model.head.weight, self.model.head.bias):
            _bump_version(t)
        self.steps += 1
        return self.grad[-1].clone()
