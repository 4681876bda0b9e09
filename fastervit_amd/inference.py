"""Library-level inference runners (replaces the eval loop plumbing of validate.py:243-344 for throughput runs).

``CompiledInference``  one GPU: the deploy plan (BN folded, 16-bit channels_last conv side on the HIP conv kernels, HAT stages on
                       the HIP kernels) run as stream shards and captured into ONE hipGraph with static buffers -- the
                       configuration ``bench.py`` measures, as a reusable object instead of caller code.
``evaluate``           a minimal eval loop with the reference's semantics (validate.py:286-344: no_grad, optional channels_last and
                       autocast, top-1 / top-5 accumulation), used by scripts/run_sharded_validate.py (one process per GPU,
                       images sharded with ``dp.shard_bounds``; no nn.DataParallel scatter / gather per step).

There is no CPU fallback: a CPU tensor raises.
"""
from __future__ import annotations

from typing import Iterable, Optional, Tuple

import torch


class CompiledInference:
    """``model.compile_inference(example)``: capture once, then ``runner(x)`` = copy-in + hipGraph replay.

    * ``example`` fixes the device, the maximum batch and the image size; shorter batches are zero-padded (images are independent in
      eval mode), other image sizes raise.
    * the returned logits are a view of the runner's static output buffer: they are overwritten by the next call (clone to keep).
    * ``join_from`` (optional): the stream shards join in front of that level and the rest of the network runs on the whole batch.
    * ``precise`` (optional): the two-term-stream conv plan (``DeployPlan.precise``).
    * a weight update after compilation is NOT picked up by the graph (its packed copies are baked in); call ``recompile()``.
    """

    def __init__(self, model, example: torch.Tensor, dtype=torch.float16, streams: int = 3, graph: bool = True, join_from=None,
                 conv_down_terms: Optional[int] = None, precise: Optional[bool] = None, slot_base: int = 0):
        if not example.is_cuda:
            raise RuntimeError("compile_inference: the example input must be on a HIP device (no CPU fallback)")
        if model.training:
            raise RuntimeError("compile_inference: call model.eval() first (inference-only path)")
        from .conv_runtime import DeployPlan
        self.model = model
        self.device = example.device
        self.plan = DeployPlan(model, dtype)
        self.plan.streams = max(1, int(streams))
        # join_from = L: the shards run levels [0, L) on their streams, join, and levels [L, end) + head run once on the whole batch
        # (DeployPlan._forward_sharded).  FasterViT-0 at batch 256: streams = 2, join_from = 3 is the measured optimum (bench.py)
        self.plan.join_from = join_from
        self.plan.slot_base = int(slot_base)   # first stage-workspace slot of this runner (PipelinedInference: one range of slots per runner in flight)
        # precise = True: two-term / fp32 streams and two-term weights on the conv side (DeployPlan.precise); with the HAT operand mode "f16x3"
        # (model.set_hat_operand_dtype) the configuration that meets logits max-abs < 1e-3 ABSOLUTE on FasterViT-4 / any-res
        # (None keeps DeployPlan's default, i.e. the FVIT_PRECISE_DEPLOY environment switch, like conv_down_terms below)
        if precise is not None:
            self.plan.precise = bool(precise)
        if conv_down_terms is not None:   # 2 = two-term weights in the Downsample.reduction convs: the accuracy option of the 16-bit plan (DeployPlan)
            self.plan.down_weight_terms = int(conv_down_terms)
        self.use_graph = bool(graph)
        self.static_x = example.detach().clone()
        self.graph = None
        self.static_y = None
        self.recompile()

    def recompile(self):
        with torch.no_grad(), torch.cuda.device(self.device):
            cur = torch.cuda.current_stream(self.device)
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(2):   # packs weights, sizes every slot's workspaces, creates the side streams -- outside the capture
                    y = self.plan.forward(self.static_x)
            cur.wait_stream(side)
            torch.cuda.synchronize(self.device)
            if not self.use_graph:
                self.graph, self.static_y = None, y
                return
            g = torch.cuda.CUDAGraph()   # a hipGraph on ROCm
            with torch.cuda.graph(g):
                self.static_y = self.plan.forward(self.static_x)
            self.graph = g
            torch.cuda.synchronize(self.device)

    @property
    def max_batch(self) -> int:
        return self.static_x.shape[0]

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda or x.device != self.device:
            raise RuntimeError(f"CompiledInference: input is on {x.device}, the runner was compiled for {self.device} (no implicit copy)")
        n = x.shape[0]
        if tuple(x.shape[1:]) != tuple(self.static_x.shape[1:]) or n > self.max_batch:
            raise RuntimeError(f"CompiledInference: input {tuple(x.shape)} does not fit the compiled shape {tuple(self.static_x.shape)}")
        with torch.no_grad(), torch.cuda.device(self.device):
            if n == self.max_batch:
                self.static_x.copy_(x, non_blocking=True)
            else:
                self.static_x[:n].copy_(x, non_blocking=True)
                self.static_x[n:].zero_()
            if self.graph is not None:
                self.graph.replay()
            else:
                self.static_y = self.plan.forward(self.static_x)
        return self.static_y[:n]


def _masked_stream(device, mask_words):
    """A HIP stream restricted to the CUs of ``mask_words`` (hipExtStreamCreateWithCUMask), wrapped for PyTorch."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    arr = (ctypes.c_uint32 * len(mask_words))(*[int(w) & 0xffffffff for w in mask_words])
    st = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(len(mask_words)), arr)
    if rc != 0 or not st.value:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {rc}")
    return torch.cuda.ExternalStream(st.value, device=device)


class PipelinedInference:
    """Throughput form of ``CompiledInference`` (r06): ``depth`` runners of the SAME compiled configuration (stream shards + join inside one hipGraph
    each), every one with its own static buffers and stage-workspace slots; step k replays runner k % depth on ITS stream, so up to ``depth`` whole-batch
    steps are in flight and the tail of step k (stage 3 on one stream, the join, the head) overlaps the stem / conv levels of step k + 1.  Every step
    still runs the whole forward on its whole batch and leaves its logits in its runner's static output; ``wait()`` joins all streams.
    Images are independent in eval mode: there is no data hazard between steps, only shared read-only packed weights."""

    def __init__(self, model, example: torch.Tensor, depth: int = 2, streams: int = 1, first: Optional[CompiledInference] = None, cu_masks=None, **kw):
        self.depth = max(1, int(depth))
        # ``first``: an existing runner of the same configuration (slot base 0) to use as runner 0
        if first is not None and getattr(first.plan, "slot_base", 0) != 0:
            raise ValueError("PipelinedInference: the runner passed as `first` must use workspace slot base 0")
        self.runners = ([first] if first is not None else []) + \
            [CompiledInference(model, example, streams=streams, slot_base=i * max(1, int(streams)), **kw) for i in range(1 if first is not None else 0, self.depth)]
        self.device = example.device
        # cu_masks (experiment, r06): one CU bit mask (list of uint32 words, bit i = CU i of the runtime's numbering) per runner: its replays go to a stream created
        # with hipExtStreamCreateWithCUMask, i.e. the steps in flight are partitioned in SPACE (each on its own CUs) instead of sharing every CU in time
        if cu_masks:
            self.streams = [_masked_stream(self.device, m) for m in cu_masks]
            if len(self.streams) != self.depth:
                raise ValueError("PipelinedInference: one CU mask per runner")
        else:
            self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.depth)]
        self._k = 0

    def set_input(self, x: torch.Tensor):
        for r in self.runners:
            r.static_x.copy_(x, non_blocking=True)
        torch.cuda.synchronize(self.device)

    def launch(self):
        """Enqueue one whole-batch step (no host sync); returns the runner whose static output will hold its logits."""
        i = self._k % self.depth
        self._k += 1
        r = self.runners[i]
        with torch.no_grad(), torch.cuda.stream(self.streams[i]):
            if r.graph is not None:
                r.graph.replay()
            else:
                r.static_y = r.plan.forward(r.static_x)
        return r

    def recompile(self):
        """Re-capture every runner (after a weight update or a change of the operand mode / plan options)."""
        self.wait()
        for r in self.runners:
            r.recompile()

    def wait(self):
        for s in self.streams:
            s.synchronize()

    def outputs(self):
        """Logits of the last step of every runner, in runner order (after ``wait``)."""
        self.wait()
        return [r.static_y for r in self.runners]


def accuracy_counts(logits: torch.Tensor, target: torch.Tensor, topk=(1, 5)) -> Tuple[int, ...]:
    """Number of samples whose target is within the top-k logits (timm.utils.accuracy counts instead of percentages, so that
    per-GPU shards can be summed exactly)."""
    k = min(max(topk), logits.shape[1])
    pred = logits.float().topk(k, dim=1).indices
    hit = pred.eq(target.view(-1, 1))
    return tuple(int(hit[:, :min(kk, k)].any(dim=1).sum().item()) for kk in topk)


@torch.no_grad()
def evaluate(model, loader: Iterable, device, amp_dtype: Optional[torch.dtype] = None, channels_last: bool = False,
             runner: Optional[CompiledInference] = None):
    """Eval loop with the semantics of the reference's validate() (validate.py:286-344): for every (input, target) batch move to the
    device (validate.py:291-293), optional channels_last (294-295), forward under ``amp_autocast`` (297-298), accumulate top-1 /
    top-5.  Returns (n_samples, top1_count, top5_count, logits of the last batch).  With ``runner`` the forward is the captured
    hipGraph (``model.compile_inference``) instead of ``model(x)``."""
    n = c1 = c5 = 0
    last = None
    for inp, tgt in loader:
        inp = inp.to(device, non_blocking=True)
        tgt = tgt.to(device, non_blocking=True)
        if channels_last:
            inp = inp.contiguous(memory_format=torch.channels_last)
        if runner is not None:
            out = runner(inp)
        elif amp_dtype is not None:
            with torch.autocast(device_type="cuda", dtype=amp_dtype):
                out = model(inp)
        else:
            out = model(inp)
        a1, a5 = accuracy_counts(out, tgt)
        n, c1, c5 = n + inp.shape[0], c1 + a1, c5 + a5
        last = out
    return n, c1, c5, last
