#!/usr/bin/env python3
"""Sharded validation: one process per GPU, each with its own slice of the dataset -- the replacement for the reference's
nn.DataParallel path (validate.py:243-244 wraps the model, :286-344 loops over batches that DataParallel scatters / gathers every
step).  No data-path collective: the only communication is ONE SUM all-reduce of (n, top-1 hits, top-5 hits) at the end, and an
optional gather of the logits to rank 0 (--gather-logits, for parity checks).

  python scripts/run_sharded_validate.py --model faster_vit_0_224 --checkpoint ck.pth.tar -b 256 --amp --channels-last
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/run_sharded_validate.py ...

Data: --synthetic N (seeded randn images + random labels, the same stream of samples whatever the number of ranks) or
--tensors FILE.pt (a dict with 'images' (N,3,H,W) and 'labels' (N,)).  Each rank evaluates samples
[shard_bounds(N, world, rank)) in batches of -b through ``model.compile_inference`` (deploy plan + stream shards in one hipGraph);
--eager runs ``model(x)`` (under autocast with --amp) instead.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="faster_vit_0_224")
    ap.add_argument("--model-kwargs", default="", help="python dict literal passed to create_model")
    ap.add_argument("--checkpoint", default="", help="state_dict file (the reference's key handling: 'state_dict' / 'model', module. prefix)")
    ap.add_argument("-b", "--batch-size", type=int, default=256)
    ap.add_argument("--synthetic", type=int, default=1024, help="number of synthetic samples (ignored with --tensors)")
    ap.add_argument("--tensors", default="", help=".pt file with {'images': (N,3,H,W), 'labels': (N,)}")
    ap.add_argument("--input-size", default="", help="HxW of the synthetic images (default: the model's pretrained_cfg)")
    ap.add_argument("--amp", action="store_true", help="validate.py --amp: fp16 autocast (the 16-bit deploy plan)")
    ap.add_argument("--channels-last", action="store_true")
    ap.add_argument("--eager", action="store_true", help="model(x) per batch instead of the captured hipGraph runner")
    ap.add_argument("--streams", type=int, default=3)
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--backend", default="", help="torch.distributed backend (default: nccl = RCCL on GPUs, gloo on cpu)")
    ap.add_argument("--gather-logits", default="", help="rank 0 writes all logits (dataset order) to this .pt file")
    ap.add_argument("--results-file", default="")
    return ap.parse_args(argv)


class ShardLoader:
    """Samples [lo, hi) of a seeded synthetic stream (sample i is the same tensor whatever the sharding) or of a tensor file."""

    def __init__(self, lo, hi, batch, size, num_classes, tensors=None):
        self.lo, self.hi, self.batch, self.size, self.num_classes, self.tensors = lo, hi, batch, size, num_classes, tensors

    def __len__(self):
        return (self.hi - self.lo + self.batch - 1) // self.batch

    def _sample(self, i):
        g = torch.Generator().manual_seed(100003 * 7 + i)
        return torch.randn((3,) + tuple(self.size), generator=g), int(torch.randint(0, self.num_classes, (1,), generator=g))

    def __iter__(self):
        for s in range(self.lo, self.hi, self.batch):
            e = min(s + self.batch, self.hi)
            if self.tensors is not None:
                yield self.tensors["images"][s:e].float(), self.tensors["labels"][s:e].long()
            else:
                xs, ys = zip(*(self._sample(i) for i in range(s, e)))
                yield torch.stack(xs), torch.tensor(ys, dtype=torch.int64)


def main(argv=None):
    args = parse(argv)
    import ast
    import fastervit_amd
    from fastervit_amd import dp
    from fastervit_amd.inference import evaluate
    rank, local_rank, world = dp.env_world()
    on_gpu = args.device.startswith("cuda")
    dist = dp.init_process_group(args.backend or ("nccl" if on_gpu else "gloo"))
    if on_gpu:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank) if on_gpu else torch.device("cpu")
    torch.manual_seed(0)
    model = fastervit_amd.create_model(args.model, **(ast.literal_eval(args.model_kwargs) if args.model_kwargs else {})).eval()
    if args.checkpoint:
        model._load_state_dict(args.checkpoint)
    model = model.to(dev)
    if args.channels_last:
        model = model.to(memory_format=torch.channels_last)
    size = tuple(int(v) for v in args.input_size.lower().split("x")) if args.input_size else tuple(model.pretrained_cfg["input_size"][-2:])
    tensors = torch.load(args.tensors) if args.tensors else None
    n_total = tensors["images"].shape[0] if tensors is not None else args.synthetic
    lo, hi = dp.shard_bounds(n_total, world, rank)
    loader = ShardLoader(lo, hi, args.batch_size, size, model.num_classes, tensors)
    runner = None
    # the hipGraph runner is the 16-bit deploy plan: it stands in for the reference's `validate.py --amp` only.  Without --amp the
    # reference evaluates in fp32, so this script does too (eager module path: fp32 conv side, HIP HAT stages)
    if on_gpu and not args.eager and args.amp:
        example = torch.zeros((args.batch_size, 3) + size, device=dev)
        if args.channels_last:
            example = example.contiguous(memory_format=torch.channels_last)
        runner = model.compile_inference(example, dtype=torch.float16, streams=args.streams)
    collected = [] if args.gather_logits else None

    def batches():
        for inp, tgt in loader:
            yield inp, tgt

    n = c1 = c5 = 0
    if collected is None:
        n, c1, c5, _ = evaluate(model, batches(), dev, amp_dtype=torch.float16 if (args.amp and on_gpu) else None,
                                channels_last=args.channels_last, runner=runner)
    else:
        for inp, tgt in batches():   # same loop, keeping the logits
            m, a1, a5, out = evaluate(model, [(inp, tgt)], dev, amp_dtype=torch.float16 if (args.amp and on_gpu) else None,
                                      channels_last=args.channels_last, runner=runner)
            n, c1, c5 = n + m, c1 + a1, c5 + a5
            collected.append(out.float().cpu().clone())
    counts = torch.tensor([n, c1, c5], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)    # the only collective: three numbers
    n_all, c1_all, c5_all = (int(v) for v in counts.tolist())
    if collected is not None:
        mine = torch.cat(collected) if collected else torch.zeros(0, model.num_classes)
        if dist is not None:
            parts = [None] * world
            dist.all_gather_object(parts, mine)     # ragged shards; small (N x num_classes fp32)
            mine = torch.cat(parts)
        if rank == 0:
            torch.save(mine, args.gather_logits)
    res = {"model": args.model, "top1": round(100.0 * c1_all / max(n_all, 1), 4), "top1_err": round(100.0 - 100.0 * c1_all / max(n_all, 1), 4),
           "top5": round(100.0 * c5_all / max(n_all, 1), 4), "top5_err": round(100.0 - 100.0 * c5_all / max(n_all, 1), 4),
           "param_count": round(sum(p.numel() for p in model.parameters()) / 1e6, 2), "img_size": size[-1], "samples": n_all,
           "world_size": world, "shard": [lo, hi], "runner": "hipGraph" if runner is not None else "eager",
           "dtype": ("float16 deploy plan (hipGraph)" if runner is not None else
                     ("float16 autocast" if (args.amp and on_gpu) else "float32 conv side" + (", HIP HAT stages (fp16 operands, fp32 accumulate)" if on_gpu else "")))}
    if rank == 0:
        print(json.dumps(res))
        if args.results_file:
            json.dump(res, open(args.results_file, "w"))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return res


if __name__ == "__main__":
    main()
