#!/usr/bin/env python3
"""bench.py -- FasterViT-0 224x224 inference throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps 50 --warmup 10
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A step is one forward pass of faster_vit_0_224 over one synthetic batch of 256 images per GPU
(BASELINE configs[1]); inputs are resident in HBM before the timed region.  Inference is
embarrassingly data parallel: every rank runs its own shard, there is no data-path collective
(SURVEY.md §8e); the only collectives are the barrier and the MAX over ranks of the elapsed time.
Rank 0 prints one JSON line.  The roofline entry is measured live with HIP events (the library's
built-in kernel timer, on the launch stream); cpu_baseline times the CPU oracle (a port of the
reference's fp32 PyTorch path) on a bounded sample on the host cores.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0   # dense bf16/fp16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="faster_vit_0_224")
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--model-kwargs", default="", help="python dict literal passed to create_model (secondary configs)")
    ap.add_argument("--input-size", default="", help="HxW override (secondary configs, e.g. 576x960)")
    ap.add_argument("--operand", default="f16", choices=["f16", "bf16"], help="MFMA operand type of the HAT kernels")
    ap.add_argument("--conv-dtype", default="f16", choices=["f16", "bf16", "f32"], help="dtype of the PyTorch-ROCm conv side")
    ap.add_argument("--mode", default="deploy", choices=["deploy", "module", "auto"],
                    help="deploy: BN folded into convs + fused glue kernels (switch_to_deploy); module: nn.Module forward under autocast")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--streams", type=int, default=3,
                    help="deploy mode: run the batch as this many shards on separate HIP streams (fork/join inside the hipGraph); "
                         "r01: 57.2k / 61.4k / 62.6k / 59.4k img/s for 1 / 2 / 3 / 4")
    ap.add_argument("--shard-sizes", type=str, default="", help="comma list of images per stream shard (default: equal split)")
    ap.add_argument("--shard-launch", choices=["free", "forkjoin"], default="forkjoin",
                    help="deploy mode with --streams > 1: 'free' = one hipGraph per shard on its own stream, replayed back to back with no "
                         "join between steps (streams drift apart, consecutive steps overlap); 'forkjoin' = one graph per step that "
                         "forks the shards and joins them (model.forward semantics; measured faster: 62.9k vs 44.9k / 62.5k / 60.0k img/s for free-running "
                         "3 / 2 / 4 streams, profiles/r01_shard_launch_sweep.log)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline sample")
    ap.add_argument("--prof-steps", type=int, default=3)
    return ap.parse_args()


def main():
    args = parse()
    from fastervit_amd import dp
    rank, local_rank, world = dp.env_world()
    dist = dp.init_process_group("nccl")  # RCCL on ROCm; None when WORLD_SIZE == 1
    assert args.gpus == world or world == 1, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import fastervit_amd
    from fastervit_amd import _lib
    torch.manual_seed(0)  # same random-init weights on every rank
    import ast
    mk = ast.literal_eval(args.model_kwargs) if args.model_kwargs else {}
    model = fastervit_amd.create_model(args.model, **mk).eval()
    sd_cpu = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(dev).to(memory_format=torch.channels_last)
    model.set_hat_operand_dtype(args.operand)
    H = W = model.pretrained_cfg["input_size"][-1]
    if args.input_size:
        H, W = (int(v) for v in args.input_size.lower().split("x"))
    gen = torch.Generator(device="cpu").manual_seed(1000 + rank)
    x_cpu = torch.randn(args.batch, 3, H, W, generator=gen)
    x = x_cpu.to(dev).contiguous(memory_format=torch.channels_last)
    conv_dt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": None}[args.conv_dtype]
    deploy = args.mode == "deploy" and conv_dt is not None
    if deploy:
        if args.shard_sizes:
            sizes = [int(v) for v in args.shard_sizes.split(",")]
            args.streams = len(sizes)
        model.switch_to_deploy(conv_dt, streams=args.streams)
        if args.shard_sizes:
            model.__dict__["_deploy_plan"].shard_sizes = sizes

    if not deploy:
        # --mode module measures the plain nn.Module path (MIOpen convs); --mode auto the same call under autocast with the
        # automatic deploy plan (what validate.py --amp gets)
        model.auto_deploy = args.mode == "auto"

    def forward(inp):
        with torch.no_grad():
            if deploy or conv_dt is None:
                return model(inp)
            with torch.autocast("cuda", dtype=conv_dt):
                return model(inp)

    # warm-up on the eager path (packs weights, allocates workspaces, lets MIOpen pick kernels)
    for _ in range(2):
        y = forward(x)
    torch.cuda.synchronize()
    graph = None
    runner = None
    if deploy and args.streams > 1 and not args.no_graph and args.shard_launch == "free":
        runner = model.__dict__["_deploy_plan"].shard_runner(x, args.streams)
    elif not args.no_graph:
        try:
            static_x = x.clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    forward(static_x)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()  # a hipGraph on ROCm
            with torch.cuda.graph(graph):
                static_y = forward(static_x)
            torch.cuda.synchronize()
        except Exception as e:  # report and measure eagerly rather than abort the bench
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); falling back to eager launches", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()

    def step():
        if runner is not None:
            runner.launch()
            return None
        if graph is not None:
            graph.replay()
            return static_y
        return forward(x)

    # W untimed + exactly K timed steps, barrier + synchronize on both sides, MAX over ranks
    elapsed = dp.timed_steps(step, args.steps, args.warmup, torch.cuda.synchronize, dist, dev)
    value = dp.whole_job_rate(args.batch * args.steps, elapsed, dist, dev)
    logits_gpu = (runner.outputs() if runner is not None else step()).float().cpu()

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3

    # ---- roofline of the dominant HAT kernel: live HIP-event timing of every launch (eager pass) ----
    if args.prof_steps <= 0:  # timeline runs under rocprofv3 (scripts/gpu_trace.sh): no eager profiling pass
        print(json.dumps({"value": round(value, 1), "ms_per_step": round(ms_per_step, 4), "roofline": None}))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    def profile_pass():
        _lib.prof_enable(True)
        for _ in range(args.prof_steps):
            forward(x)
        torch.cuda.synchronize()
        pr = _lib.prof_collect()
        _lib.prof_enable(False)
        return pr

    def roofline_of(pr, dom=None):
        # dominant HAT kernel = the MFMA kernel family with the largest summed time per step; its roof follows from its
        # algorithmic intensity (FLOP per compulsory HBM byte) against the ridge 2.5e15 / 8e12 = 312 FLOP/B
        if dom is None:
            kinds = [k for k in pr if (k.startswith("gemm") or k in ("mlp_fused", "attn_block_fused")) and pr[k]["launches"]]
            dom = max(kinds, key=lambda k: pr[k]["ms"])
        e = pr[dom]
        sec = e["ms"] * 1e-3
        intensity = e["flops"] / max(e["bytes"], 1.0)
        if intensity >= MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9):
            bound, achieved, peak, unit = "mfma", e["flops"] / sec / 1e12, MFMA_PEAK_TFLOPS, "TFLOP/s"
        else:
            bound, achieved, peak, unit = "hbm", e["bytes"] / sec / 1e9, HBM_PEAK_GBS, "GB/s"
        return dom, {"kernel": f"{dom} <{args.operand}>", "bound": bound, "achieved": round(achieved, 2), "peak": peak, "unit": unit,
                     "frac": round(achieved / peak, 4), "traffic": None,
                     "flop_per_byte": round(intensity, 1), "tflops": round(e["flops"] / sec / 1e12, 2),
                     "launches_per_step": e["launches"] // args.prof_steps,
                     "avg_launch_us": round(e["ms"] * 1e3 / e["launches"], 2),
                     "algorithmic_gflop_per_launch": round(e["flops"] / e["launches"] / 1e9, 3),
                     "algorithmic_mbyte_per_launch": round(e["bytes"] / e["launches"] / 1e6, 3)}

    def pmc_traffic(kind, shard_sized):
        """HBM bytes per launch of the kernel family `kind` from the committed rocprofv3 PMC passes (profiles/, collected with
        scripts/gpu_prof.sh + scripts/pmc_traffic_summary.py on the same command in eager mode): bench.py cannot sample PMCs on
        itself.  Picks the (kernel, grid) row with the largest summed time, i.e. the launch shape that dominates the family."""
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_hbm_traffic_by_kernel_v13.json")
        prefix = {"mlp_fused": "mlp_fused_kernel", "attn_block_fused": "attnblk_kernel", "gemm_bias": f"gemm_kernel<{args.operand},0",
                  "gemm_gelu": f"gemm_kernel<{args.operand},1", "gemm_residual": f"gemm_kernel<{args.operand},2"}.get(kind)
        if not shard_sized or prefix is None or not os.path.exists(path) or args.model != "faster_vit_0_224" or args.batch != 256:
            return None, None
        rows = [r for r in json.load(open(path))["kernels"] if r["kernel"].startswith(prefix)]
        if not rows:
            return None, None
        r = max(rows, key=lambda r: r["total_us"])
        return int(r["hbm_traffic_mb"] * 1e6), (f"profiles/r01_pmc_hbm_traffic_by_kernel_v13.json: {r['kernel']} x {r['workgroups']} workgroups, "
                                                f"read {r['hbm_read_mb']} MB (2 x FETCH_SIZE) + write {r['hbm_write_mb']} MB per launch")

    def kernel_table(pr):
        return {k: {"launches_per_step": v["launches"] // args.prof_steps, "ms_per_step": round(v["ms"] / args.prof_steps, 4),
                    "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2), "gbs": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1)}
                for k, v in pr.items() if v["launches"]}

    # ---- roofline of the dominant HAT kernel: live HIP-event timing of every launch in an eager pass with the SAME launches as
    # the timed region (shard-sized), the shards issued one after the other on one stream: an event pair then brackets one
    # kernel alone on the GPU -- the regime rocprofv3 --kernel-trace measures too (it serialises dispatches), so the two agree ----
    plan = model.__dict__.get("_deploy_plan")
    if plan is not None:
        plan.serialize_shards = True
    prof = profile_pass()
    if plan is not None:
        plan.serialize_shards = False
    hat_ms = sum(e["ms"] for k, e in prof.items() if k not in ("other", "conv3x3")) / args.prof_steps
    dom, roofline = roofline_of(prof)
    roofline["traffic"], roofline["traffic_source"] = pmc_traffic(dom, deploy and args.streams == 3)
    kernels = kernel_table(prof)
    # ---- the same kernels with the GPU to themselves: one stream, whole-batch launches (kernel quality, not job throughput) ----
    roofline_isolated = None
    if plan is not None and getattr(plan, "streams", 1) > 1:
        shards = plan.streams
        plan.streams = 1
        forward(x)  # sizes the whole-batch workspace outside the profiled pass
        torch.cuda.synchronize()
        prof1 = profile_pass()
        plan.streams = shards
        _, roofline_isolated = roofline_of(prof1, dom)
        roofline_isolated["launch"] = "eager, 1 stream, whole-batch launches"
        roofline_isolated["kernels"] = kernel_table(prof1)

    # ---- CPU baseline: the oracle (port of the reference fp32 CPU path) on a bounded sample ----
    cpu = None
    parity = None
    if not args.no_cpu_baseline and world == 1:  # reported at N = 1 only (rank 0's host cores are shared with the other ranks otherwise)
        from oracle.model_reference import model_forward
        from tests.cases import CASES
        arch = dict(CASES["fvit0_224"]["arch"]) if args.model == "faster_vit_0_224" else None
        if arch is not None:
            nb = 8
            xs = x_cpu[:nb]
            ncpu = os.cpu_count() or 1
            # small-batch fp32 inference does not scale to hundreds of threads: pick the fastest of a few
            # thread counts on one iteration each, then time the bounded sample with that count
            cands = sorted({t for t in (8, 16, 32, 64) if t <= ncpu} or {ncpu})
            torch.set_num_threads(cands[0])
            ref = model_forward(sd_cpu, xs, arch)  # warm-up + parity reference
            best_t, best_dt = cands[0], float("inf")
            for t in cands:
                torch.set_num_threads(t)
                model_forward(sd_cpu, xs, arch)
                t1 = time.perf_counter()
                model_forward(sd_cpu, xs, arch)
                dt = time.perf_counter() - t1
                if dt < best_dt:
                    best_t, best_dt = t, dt
            torch.set_num_threads(best_t)
            n, t_cpu = 0, 0.0
            while t_cpu < args.cpu_seconds and n < 50:
                t1 = time.perf_counter()
                model_forward(sd_cpu, xs, arch)
                t_cpu += time.perf_counter() - t1
                n += 1
            cpu = {"value": round(nb * n / t_cpu, 2), "unit": "images/s", "cores": best_t, "kind": "port",
                   "sample": f"{args.model} fp32 oracle (port of the reference CPU path), batch {nb}, {n} iterations "
                             f"({t_cpu:.1f} s), {best_t} of {ncpu} host threads (fastest of {cands})"}
            err = (logits_gpu[:nb] - ref).abs().max().item()
            parity = {"logits_max_abs_err": float(f"{err:.3e}"), "logits_abs_max": round(ref.abs().max().item(), 4),
                      "vs": "CPU oracle fp32, first 8 images of rank 0's batch", "weights": "random init (seed 0)"}

    out = {
        "metric": ("images/sec FasterViT-0 224x224 inference, bs=256/GPU" if (args.model, args.batch, H) == ("faster_vit_0_224", 256, 224)
                   else f"images/sec {args.model} {H}x{W} inference, bs={args.batch}/GPU (secondary config)"), "value": round(value, 1), "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.operand, "data": "synthetic",
        "config": {"workload": f"{args.model} inference, {H}x{W}, batch {args.batch}/GPU, random-init weights",
                   "global_batch": args.batch * world, "parallelism": f"dp{world} (independent shards, no data-path collective)",
                   "hat_operands": args.operand, "conv_side": (f"deploy plan: BN folded, {args.conv_dtype} channels_last, fused HIP conv3x3 (halo-tiled / implicit-GEMM) + stem + LayerNorm2d "
                                               "kernels (MIOpen only for channel counts the kernels do not cover; none in this model)"
                                 if deploy else (f"model(x) under autocast {args.conv_dtype}: automatic deploy plan ({model.auto_deploy_streams} stream shards)" if args.mode == "auto"
                                                 else f"PyTorch-ROCm nn.Module forward, channels_last, autocast {args.conv_dtype}")),
                   "launch": (f"{args.streams} free-running stream shards, one hipGraph replay per shard and step, no join between steps" if runner is not None
                              else ("hipGraph replay" if graph is not None else "eager") + (f", {args.streams} stream shards (fork/join)" if deploy and args.streams > 1 else ""))},
        "roofline": roofline, "roofline_isolated": roofline_isolated, "cpu_baseline": cpu, "parity": parity,
        "hat_ms_per_step": round(hat_ms, 4), "hat_kernels": kernels,
    }
    print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
