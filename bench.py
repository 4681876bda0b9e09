#!/usr/bin/env python3
"""bench.py -- FasterViT-0 224x224 inference throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps 50 --warmup 10
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A step is one forward pass of faster_vit_0_224 over one synthetic batch of 256 images per GPU (BASELINE configs[1]); inputs are
resident in HBM before the timed region.  Inference is embarrassingly data parallel: every rank runs its own shard, there is no
data-path collective; the only collectives are the barrier and the MAX / SUM reductions that turn per-rank
timings into one whole-job figure.  Rank 0 prints ONE COMPACT JSON line (< 8 KB: the driver parses the last stdout line; r03's 26 KB
line was cut by its stdout tail) and writes the full record to gpurun_out/bench_detail.json (untracked; --record PATH for a committed copy):

  value / ms_per_step     W untimed + exactly K timed hipGraph replays, barrier + synchronize on both sides, MAX over ranks.  The graph
                          holds the deploy plan as 2 stream shards through level 2, joined, level 3 + head on the whole batch (--streams,
                          --join-from; fastervit_amd.inference.CompiledInference)
  step_ms                 min / median / max of individually event-timed steps (a separate pass; dispersion of the number above)
  roofline                the DOMINANT KERNEL BY TIME: launches of the timed configuration are summed per kernel name (the key of a
                          rocprofv3 --kernel-trace --stats row, conv kernels included, no weighting); the kernel with the largest
                          summed time per step is reported through its heaviest launch shape: live HIP-event duration per launch
                          (library kernel timer, on the launch stream), algorithmic bytes / FLOPs of THAT shape; beside it the duration of
                          the same (kernel, workgroups) in the committed rocprofv3 kernel trace (avg_launch_us_rocprof, frac_rocprof) and
                          its PMC traffic from the committed FETCH_SIZE / WRITE_SIZE passes (null when the committed file has no such row)
  parity / parity_<op>    logits max-abs error vs the fp32 CPU oracle over ALL images of the timed batch (r06; worst_image = its index), on synthetic weights of the
                          'init' family of tests/synth.py (reference init + gamma ~ U(0.5, 1.5), BN statistics, biases).  The timed
                          operand type first; bf16, f16x2, bf16x2 beside it with their own error AND their own images/s
  secondary               BASELINE configs 3 and 5 (faster_vit_4_224 bs 128; faster_vit_4_any_res 576x960 bs 8), each TIMED and CHECKED ON THE
                          SAME 8 IMAGES in two configurations: the PRECISE deploy plan (two-term conv streams + HAT operands f16x3: meets
                          north_star's ABSOLUTE logits max-abs < 1e-3 on models whose logits reach |7|) = value / parity, and `fast` (the 16-bit
                          deploy plan: relative claim only)   (N = 1 only)
  cpu_baseline            the CPU oracle (kind "port": a restatement of the reference's fp32 PyTorch path pinned by golden vectors generated
                          from the real reference; /root/reference does not exist on the GPU box) on the host cores, batch 8 and batch 64
"""
import argparse
import ast
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0   # dense bf16/fp16, /opt/skills/guides/MI355X_MICROARCH.md
N_CU = 256                 # compute units of an MI355X (8 XCDs x 32)
HBM_PEAK_GBS = 8000.0
RIDGE = MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
OPERAND_MODES = ("f16", "bf16", "f16x2", "bf16x2")   # the modes the headline configuration is timed in (parity_<mode> + images/s)
ALL_OPERAND_MODES = OPERAND_MODES + ("f16x3", "bf16x3")   # fastervit_amd.hat_runtime.OPERAND_MODES
WEIGHT_SEED = 1234          # tests/cases.py SEED: the weights of the committed golden fixtures
# scripts/gpu_pmc_traffic.sh -> scripts/pmc_traffic_summary.py; the newest committed round wins
ROUNDS = (6, 5, 4, 3, 2)
PMC_FILE = next((f for f in (os.path.join("profiles", f"r0{r}_pmc_hbm_traffic_by_kernel.json") for r in ROUNDS)
                 if os.path.exists(os.path.join(ROOT, f))), os.path.join("profiles", "r02_pmc_hbm_traffic_by_kernel.json"))
# rocprofv3 --kernel-trace --stats of this command, per (kernel, launch shape) (scripts/summarize_rocprof_db.py): the newest committed round
ROCPROF_SHAPES = next((f for f in (os.path.join("profiles", f"r0{r}_bench_final_fvit_kernels_by_shape.csv") for r in ROUNDS)
                       if os.path.exists(os.path.join(ROOT, f))), None)
EVENT_VS_ROCPROF_TOL = 0.25   # the live event timer and the committed kernel trace must agree within this (else roofline.timer_mismatch is set)
OUTLIER = 1.5                 # a launch slower than OUTLIER x the median of its (kernel, shape) is dropped from the average (and counted)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="faster_vit_0_224")
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--model-kwargs", default="", help="python dict literal passed to create_model (secondary configs)")
    ap.add_argument("--input-size", default="", help="HxW override (secondary configs, e.g. 576x960)")
    ap.add_argument("--operand", default="f16", choices=list(ALL_OPERAND_MODES),
                    help="operand mode of the HAT kernels: f16 / bf16 (16-bit operands rounded once), f16x2 / bf16x2 (weights as two terms hi + lo), f16x3 / bf16x3 (weights AND activations as two terms)")
    ap.add_argument("--conv-dtype", default="f16", choices=["f16", "bf16", "f32"], help="dtype of the conv side")
    ap.add_argument("--mode", default="deploy", choices=["deploy", "module", "auto"],
                    help="deploy: BN folded into convs + fused HIP conv kernels (model.compile_inference); module: nn.Module forward under "
                         "autocast; auto: model(x) under autocast (automatic deploy plan)")
    ap.add_argument("--precise", action="store_true",
                    help="deploy mode: the two-term-stream conv plan (DeployPlan.precise); with --operand f16x3 the configuration that meets the ABSOLUTE "
                         "1e-3 bar on FasterViT-4 / any-res (the timed configuration of the secondary entries)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--streams", type=int, default=1, help="deploy mode: stream shards of the batch (fork / join inside the hipGraph); r05 default: 2 with --join-from 3 and --inflight 1")
    ap.add_argument("--shard-sizes", type=str, default="", help="comma list of images per stream shard (default: equal split)")
    ap.add_argument("--shard-launch", choices=["free", "forkjoin"], default="forkjoin",
                    help="'free' = one hipGraph per shard on its own stream, no join between steps (r01: slower); 'forkjoin' = one graph per step")
    ap.add_argument("--inflight", type=int, default=2,
                    help="deploy + hipGraph: whole-batch steps in flight (fastervit_amd.inference.PipelinedInference: step k replays runner k %% N on its own stream, "
                         "so the tail of a step overlaps the front of the next one; 1 = one graph, steps strictly one after the other).  Default since r06: 2 steps in flight, "
                         "whole-batch launches (--streams 1): +4..8 %% images/s over r05's 2 stream shards + join inside one graph, profiles/r06_steps_in_flight_ab.log")
    ap.add_argument("--cu-mask", default="", choices=["", "halves", "interleaved", "xcd"],
                    help="experiment: the steps in flight on CU-masked streams (hipExtStreamCreateWithCUMask): halves = CU bits [0,128) / [128,256); interleaved = even / odd bits; "
                         "xcd = bits with (i % 8) < 4 / >= 4")
    ap.add_argument("--join-from", type=int, default=0,
                    help="deploy mode: shards run levels [0, L) on their streams, join, levels [L, end) run once on the whole batch (0: off). "
                         "r04 / r05 default: 3 with 2 shards ( A/B in one box x 3: 82.2-83.2k vs 80.2-81.8k images/s for 3 shards without the join: the last "
                         "stage of FasterViT-0 is one 49-token window per image, 86-workgroup launches per shard)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of each cpu_baseline sample")
    ap.add_argument("--prof-steps", type=int, default=3, help="eager HIP-event passes for the roofline rows (0: skip)")
    ap.add_argument("--no-secondary", action="store_true", help="skip BASELINE configs 3 and 5")
    ap.add_argument("--no-train-step", action="store_true", help="skip the training-step throughput entry (faster_vit_0_224, batch 64; N = 1 only)")
    ap.add_argument("--no-rank-parity", action="store_true", help="N > 1: skip the per-rank parity check against the CPU oracle")
    ap.add_argument("--secondary-steps", type=int, default=10)
    ap.add_argument("--secondary-streams", type=int, default=0, help="stream shards of the secondary configs' precise plan (0: the default, 2)")
    ap.add_argument("--no-modes", action="store_true", help="skip the parity / images-per-second legs of the other operand modes")
    ap.add_argument("--record", default="", help="also write the full record to this path (e.g. profiles/r05_bench_final_detail.json); the default "
                                                 "run writes only the untracked gpurun_out/bench_detail.json and leaves the work tree clean")
    ap.add_argument("--launch-selftest", action="store_true",
                    help="exercise ONLY the rank launch / barrier / reduction path on CPU (gloo), no model: prints n_gpus = ranks that ran")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------------------------------
# one configuration: build, warm up, capture, time
# ------------------------------------------------------------------------------------------------------------------------
class Config:
    def __init__(self, args, dev, rank, model_name, batch, hw=None, model_kwargs=None, streams=3):
        import fastervit_amd
        self.args, self.dev, self.name, self.batch = args, dev, model_name, batch
        torch.manual_seed(0)
        self.model = fastervit_amd.create_model(model_name, **(model_kwargs or {})).eval()
        # synthetic weights keyed by state_dict name, identical on every rank: the reference's init plus seeded gamma ~ U(0.5, 1.5),
        # BN running statistics and biases (tests/synth.py 'init' family, SURVEY.md §4 trap / §8d) -- with the plain init the
        # layer-scale gammas of FasterViT-4 are 1e-5 and a parity check would see the conv side only
        from tests.synth import synth_state_dict
        self.model.load_state_dict(synth_state_dict(self.model.state_dict(), seed=WEIGHT_SEED, family="init"))
        self.sd_cpu = {k: v.clone() for k, v in self.model.state_dict().items()}
        self.model = self.model.to(dev).to(memory_format=torch.channels_last)
        self.model.set_hat_operand_dtype(args.operand)
        H = W = self.model.pretrained_cfg["input_size"][-1]
        if hw:
            H, W = hw
        self.H, self.W = H, W
        gen = torch.Generator(device="cpu").manual_seed(1000 + rank)
        self.x_cpu = torch.randn(batch, 3, H, W, generator=gen)
        self.x = self.x_cpu.to(dev).contiguous(memory_format=torch.channels_last)
        self.conv_dt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": None}[args.conv_dtype]
        self.deploy = args.mode == "deploy" and self.conv_dt is not None
        self.streams = streams if self.deploy else 1
        self.runner = self.graph = self.static_y = self.free_runner = None
        self.plan = None

    def eager(self, inp):
        with torch.no_grad():
            if self.deploy:
                return self.plan.forward(inp)
            if self.conv_dt is None:
                return self.model(inp)
            with torch.autocast("cuda", dtype=self.conv_dt):
                return self.model(inp)

    def prepare(self):
        a = self.args
        if self.deploy:
            # the library-level runner: deploy plan + stream shards + ONE hipGraph with static buffers
            jf = a.join_from if (a.join_from > 0 and self.name == a.model and self.streams > 1) else None
            self.runner = self.model.compile_inference(self.x, dtype=self.conv_dt, streams=self.streams, graph=not a.no_graph, join_from=jf,
                                                       precise=bool(getattr(a, "precise", False)))
            self.plan = self.runner.plan
            # (equal shards: 132 + 124 measured +0.5 % in one box, scripts/r04_calls/call8.sh, but puts the 132-image shard's attnblk launch at 528
            # workgroups = two rounds of the 512 resident slots, 74 vs 50 us; not adopted)
            sizes = a.shard_sizes
            if sizes:
                self.plan.shard_sizes = [int(v) for v in sizes.split(",")]
                self.plan.streams = len(self.plan.shard_sizes)
                self.runner.recompile()
            if self.streams > 1 and not a.no_graph and a.shard_launch == "free":
                self.free_runner = self.plan.shard_runner(self.x, self.streams)
            self.pipe = None
            if getattr(a, "inflight", 1) > 1 and not a.no_graph and not sizes and self.free_runner is None:
                from fastervit_amd.inference import PipelinedInference
                masks = None
                if getattr(a, "cu_mask", "") and a.inflight == 2:
                    def words(pred):
                        return [sum((1 << b) for b in range(32) if pred(32 * w + b)) for w in range(8)]
                    sel = {"halves": lambda i: i < 128, "interleaved": lambda i: i % 2 == 0, "xcd": lambda i: (i % 8) < 4}[a.cu_mask]
                    masks = [words(sel), words(lambda i: not sel(i))]
                self.pipe = PipelinedInference(self.model, self.x, depth=a.inflight, streams=self.streams, first=self.runner, cu_masks=masks, dtype=self.conv_dt,
                                               join_from=jf, precise=bool(getattr(a, "precise", False)))
            return
        self.model.auto_deploy = a.mode == "auto"
        for _ in range(2):
            self.eager(self.x)
        torch.cuda.synchronize()
        if not a.no_graph:
            static_x = self.x.clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    self.eager(static_x)
            torch.cuda.current_stream().wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.static_y = self.eager(static_x)
            torch.cuda.synchronize()

    def set_plan(self, **attrs):
        """Plan options on every runner's plan (the pipelined form has one plan per runner)."""
        plans = [r.plan for r in self.pipe.runners] if getattr(self, "pipe", None) is not None else ([self.plan] if self.plan is not None else [])
        for pl in plans:
            for k, v in attrs.items():
                setattr(pl, k, v)

    def recompile(self):
        if getattr(self, "pipe", None) is not None:
            self.pipe.recompile()
        elif self.runner is not None:
            self.runner.recompile()

    def step(self):
        if getattr(self, "pipe", None) is not None:
            return self.pipe.launch().static_y    # (valid after the next sync; the timed region syncs at its end)
        if self.free_runner is not None:
            self.free_runner.launch()
            return None
        if self.runner is not None:
            if self.runner.graph is not None:
                self.runner.graph.replay()     # inputs already resident in the static buffer (HBM) before the timed region
                return self.runner.static_y
            self.runner.static_y = self.plan.forward(self.runner.static_x)
            return self.runner.static_y
        if self.graph is not None:
            self.graph.replay()
            return self.static_y
        return self.eager(self.x)

    def logits(self):
        if getattr(self, "pipe", None) is not None:
            # every runner of the pipeline is checked: one more step each, then the element-wise WORST logits error is what parity sees
            for _ in range(self.pipe.depth):
                self.pipe.launch()
            outs = [o.float().cpu() for o in self.pipe.outputs()]
            # the runners see the same input and run the same kernels: bitwise equal (tests/test_gpu_runtime.py asserts it); here the difference is RECORDED
            # (runners_max_abs_diff in the line) and parity is taken on the runner that is worst against runner 0, so a mismatch can only make parity worse
            diffs = [(o - outs[0]).abs().nan_to_num(nan=float("inf")).max().item() for o in outs]
            self.runners_max_abs_diff = max(diffs)
            return outs[max(range(len(outs)), key=lambda i: diffs[i])]
        out = self.free_runner.outputs() if self.free_runner is not None else self.step()
        torch.cuda.synchronize()
        return out.float().cpu()

    def launch_desc(self):
        if getattr(self, "pipe", None) is not None:
            jf = getattr(self.plan, "join_from", None)
            return (f"hipGraph replay, {self.pipe.depth} whole-batch steps in flight (step k replays runner k % {self.pipe.depth} on its own stream; "
                    f"fastervit_amd.inference.PipelinedInference), each {self.streams} stream shards (fork/join inside the graph)" +
                    (f", level {jf}+ joined on the whole batch" if jf else ""))
        if self.free_runner is not None:
            return f"{self.streams} free-running stream shards, one hipGraph replay per shard and step"
        g = "eager" if (self.args.no_graph or (self.runner is None and self.graph is None)) else "hipGraph replay"
        jf = getattr(self.plan, "join_from", None) if self.plan is not None else None
        return g + (f", {self.streams} stream shards (fork/join inside the graph; fastervit_amd.inference.CompiledInference)" if self.deploy and self.streams > 1 else "") + \
            (f", level {jf}+ joined on the whole batch" if jf and self.deploy and self.streams > 1 else "")


def step_dispersion(cfg, n):
    """min / median / max over n individually timed steps (event pair per step; not the throughput measurement)."""
    pipe = getattr(cfg, "pipe", None)
    if pipe is not None:
        # steps in flight: a step's LATENCY is the time from its first to its last kernel on its runner's stream, with the other steps' kernels beside it
        # (about `depth` x the throughput step time); the event pair is recorded on that stream
        evs = []
        torch.cuda.synchronize()
        for _ in range(n):
            i = pipe._k % pipe.depth
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(pipe.streams[i]):
                a.record()
            cfg.step()
            with torch.cuda.stream(pipe.streams[i]):
                b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in evs)
        return {"n": n, "min": round(ms[0], 4), "median": round(statistics.median(ms), 4), "max": round(ms[-1], 4),
                "note": f"LATENCY of a step with {pipe.depth} steps in flight (event pair on the step's own stream); throughput = ms_per_step"}
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    torch.cuda.synchronize()
    for a, b in evs:
        a.record()
        cfg.step()
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    return {"n": n, "min": round(ms[0], 4), "median": round(statistics.median(ms), 4), "max": round(ms[-1], 4),
            "note": "one event pair per step, steps issued back to back (includes the replay launch gap)"}


# ------------------------------------------------------------------------------------------------------------------------
# roofline rows per (kernel, launch shape)
# ------------------------------------------------------------------------------------------------------------------------
def profile_shapes(cfg, prof_steps, serialize=True):
    """Eager pass with the SAME launches as the timed region (shard-sized), shards issued one after the other on one stream, an
    event pair around every launch: a kernel's duration is then its own (the regime rocprofv3 --kernel-trace measures too)."""
    from fastervit_amd import _lib
    plan = cfg.plan
    if plan is not None:
        plan.serialize_shards = serialize
    cfg.eager(cfg.x)   # sizes anything the serialized order needs, outside the recorded pass
    torch.cuda.synchronize()
    _lib.prof_enable(True)
    for _ in range(prof_steps):
        cfg.eager(cfg.x)
    torch.cuda.synchronize()
    recs = _lib.prof_records()
    _lib.prof_enable(False)
    if plan is not None:
        plan.serialize_shards = False
    return summarize_launches(recs, prof_steps)


def summarize_launches(recs, prof_steps):
    """Per (kernel, launch shape) rows from the library's launch records [{kind, name, grid, flops, bytes, ms}] of prof_steps passes."""
    rows = {}
    for r in recs:
        key = (r["kind"], r["name"], r["grid"], round(r["flops"]), round(r["bytes"]))
        e = rows.setdefault(key, dict(kind=r["kind"], kernel=r["name"] or r["kind"], workgroups=r["grid"], launches=0, samples=[],
                                      flops=r["flops"], bytes=r["bytes"]))
        e["launches"] += 1
        e["samples"].append(r["ms"])
    out = []
    for e in rows.values():
        # robust per-launch duration (r05): an event pair occasionally brackets a stall that is not the kernel's (first pass after a
        # plan change, a clock ramp): in the r04 driver run one such launch moved ctblk8's MEAN from 34 to 58 us and with it the dominant-kernel
        # selection.  Launches slower than OUTLIER x the median of their own (kernel, shape) are dropped and counted; the rest are averaged.
        med = statistics.median(e["samples"])
        kept = [v for v in e["samples"] if v <= OUTLIER * med] or [med]
        us = sum(kept) / len(kept) * 1e3
        per_step = max(1, e["launches"] // max(prof_steps, 1))
        inten = e["flops"] / max(e["bytes"], 1.0)
        bound = "mfma" if inten >= RIDGE else "hbm"
        tf, gbs = e["flops"] / us / 1e6, e["bytes"] / us / 1e3
        out.append(dict(kernel=e["kernel"], kind=e["kind"], workgroups=e["workgroups"], launches_per_step=per_step,
                        avg_launch_us=round(us, 2), median_launch_us=round(med * 1e3, 2), outlier_launches_dropped=len(e["samples"]) - len(kept),
                        ms_per_step=round(us * per_step / 1e3, 4),
                        algorithmic_mflop_per_launch=round(e["flops"] / 1e6, 2), algorithmic_mbyte_per_launch=round(e["bytes"] / 1e6, 3),
                        flop_per_byte=round(inten, 1), bound=bound, tflops=round(tf, 2), gbs=round(gbs, 1),
                        frac=round((tf / MFMA_PEAK_TFLOPS) if bound == "mfma" else (gbs / HBM_PEAK_GBS), 4)))
    out.sort(key=lambda r: -r["ms_per_step"])
    return out


def shapes_cu(shapes):
    for r in shapes:
        r["cu_share"] = round(min(r["workgroups"], N_CU) / N_CU, 3)
        r["gpu_ms_per_step"] = round(r["ms_per_step"] * r["cu_share"], 4)
    return shapes


def pmc_file_for(model_name):
    """Committed PMC summary of a configuration: the headline file, or profiles/r03_pmc_hbm_traffic_by_kernel_<model>.json."""
    if model_name in (None, "faster_vit_0_224"):
        return PMC_FILE
    cands = [os.path.join("profiles", f"r0{r}_pmc_hbm_traffic_by_kernel_{model_name}.json") for r in ROUNDS]   # the newest committed round wins
    return next((f for f in cands if os.path.exists(os.path.join(ROOT, f))), cands[-1])


def pmc_row(kernel, workgroups, pmc_file=None):
    """HBM bytes per launch of (kernel family, workgroups) from the committed rocprofv3 PMC passes of THIS command in eager mode
    (bench.py cannot sample PMCs on itself); None when the committed file has no row of that launch shape."""
    path = os.path.join(ROOT, pmc_file or PMC_FILE)
    if not os.path.exists(path):
        return None
    fam = kernel.split("<")[0].split(" ")[0]
    best = None
    for r in json.load(open(path)).get("kernels", []):
        if r["kernel"].startswith(fam) and int(r.get("workgroups", -1)) == int(workgroups):
            if kernel.startswith(("gemm_kernel<", "gemm_pp_kernel<")):   # epilogue id is the 2nd template argument of gemm[_pp]_kernel<T,EPI,...>
                targs = r["kernel"].split("<", 1)[1].split(",")
                if len(targs) < 2 or targs[1].strip().rstrip(">") != kernel.split("<", 1)[1][0]:
                    continue
            if best is None or r["total_us"] > best["total_us"]:
                best = r
    return best


def rocprof_row(kernel, workgroups):
    """Average duration of (kernel family, workgroups) in the committed rocprofv3 kernel trace of this command (None: no such row).  The live HIP-event
    timer and the kernel trace differ by ~10 % on short kernels (the event pair sees launch gaps rocprofv3 does not, the trace runs a slightly slower
    clock): both are reported."""
    if not ROCPROF_SHAPES:
        return None
    import csv
    fam = kernel.split("<")[0].split(" ")[0]
    best = None
    try:
        with open(os.path.join(ROOT, ROCPROF_SHAPES)) as f:
            for r in csv.DictReader(f):
                if r["Name"].startswith(fam) and int(r["Workgroups"]) == int(workgroups):
                    if best is None or float(r["TotalUs"]) > float(best["TotalUs"]):
                        best = r
    except (OSError, KeyError, ValueError):
        return None
    return None if best is None else {"avg_us": round(float(best["AverageUs"]), 2), "calls": int(best["Calls"]), "file": ROCPROF_SHAPES, "name": best["Name"]}


def dominant_by_time(shapes):
    """The dominant kernel BY TIME: launches summed per kernel name -- the key of a rocprofv3 --kernel-trace --stats row; every kernel
    with algorithmic work counts, conv kernels included, no weighting -- and, inside that kernel, its heaviest launch shape."""
    fam = {}
    for r in shapes:
        if r["algorithmic_mflop_per_launch"] > 0 or r["algorithmic_mbyte_per_launch"] > 0:
            if r["kind"] != "other":
                fam.setdefault(r["kernel"], []).append(r)
    if not fam:
        return None, 0.0
    name = max(fam, key=lambda k: sum(r["ms_per_step"] for r in fam[k]))
    return max(fam[name], key=lambda r: r["ms_per_step"]), round(sum(r["ms_per_step"] for r in fam[name]), 4)


def kernel_family(name):
    """gemm_kernel<..> and gemm_pp_kernel<..> of every epilogue are ONE family (the Linear layers); conv kernels another; else the kernel name's stem."""
    stem = name.split("<")[0].split(" ")[0]
    if stem.startswith("gemm"):
        return "gemm (Linear layers: gemm_kernel + gemm_pp_kernel, all epilogues)"
    if stem.startswith(("conv3x3", "stem")):
        return "conv (implicit-GEMM 3x3 convs + stem)"
    return stem


def dominant_family(shapes, terms_mult=None):
    """r06: the dominant kernel FAMILY by summed time, its time-weighted fraction of the MFMA peak on ALGORITHMIC FLOPs, and the
    MFMA ISSUE fraction beside it: in the two-term plans a Linear layer issues 3 (x3: hi.hi + hi.lo + lo.hi) and a conv 2-3 MFMA passes per algorithmic FLOP."""
    fam = {}
    for r in shapes:
        if r["kind"] == "other" or (r["algorithmic_mflop_per_launch"] <= 0 and r["algorithmic_mbyte_per_launch"] <= 0):
            continue
        f = fam.setdefault(kernel_family(r["kernel"]), {"ms": 0.0, "mflop": 0.0, "mbyte": 0.0})
        f["ms"] += r["ms_per_step"]
        f["mflop"] += r["algorithmic_mflop_per_launch"] * r["launches_per_step"]
        f["mbyte"] += r["algorithmic_mbyte_per_launch"] * r["launches_per_step"]
    if not fam:
        return None
    total = sum(f["ms"] for f in fam.values())
    name = max(fam, key=lambda k: fam[k]["ms"])
    f = fam[name]
    tf = f["mflop"] / max(f["ms"], 1e-9) / 1e3   # MFLOP / ms = GFLOP/s -> / 1e3 = TFLOP/s
    mult = (terms_mult or {}).get("gemm" if name.startswith("gemm") else "conv" if name.startswith("conv") else "other", 1.0)
    return {"family": name, "ms_per_step_serialized": round(f["ms"], 4), "share_of_kernel_time": round(f["ms"] / max(total, 1e-9), 3),
            "algorithmic_tflops": round(tf, 1), "frac": round(tf / MFMA_PEAK_TFLOPS, 4),
            "mfma_issue_mult": mult, "mfma_issue_frac": round(tf * mult / MFMA_PEAK_TFLOPS, 4),
            "note": "time-weighted over the family's launches of one serialized pass; frac counts algorithmic FLOPs, mfma_issue_frac the MFMA passes issued"}


def roofline_entry(row, operand, family_ms=None, pmc_file=None):
    if row is None:
        return None
    pmc_file = pmc_file or PMC_FILE
    e = {"kernel": f"{row['kernel']} <{operand}> x {row['workgroups']} workgroups", "bound": row["bound"],
         "achieved": row["tflops"] if row["bound"] == "mfma" else row["gbs"], "peak": MFMA_PEAK_TFLOPS if row["bound"] == "mfma" else HBM_PEAK_GBS,
         "unit": "TFLOP/s" if row["bound"] == "mfma" else "GB/s", "frac": row["frac"], "traffic": None, "traffic_source": None,
         "flop_per_byte": row["flop_per_byte"], "tflops": row["tflops"], "gbs": row["gbs"], "launches_per_step": row["launches_per_step"],
         "avg_launch_us": row["avg_launch_us"], "ms_per_step": row["ms_per_step"],
         "algorithmic_mflop_per_launch": row["algorithmic_mflop_per_launch"], "algorithmic_mbyte_per_launch": row["algorithmic_mbyte_per_launch"],
         "kernel_ms_per_step_all_shapes": family_ms,
         "cu_share": row.get("cu_share"),
         "frac_of_occupied_cus": round(row["frac"] / max(row.get("cu_share") or 1.0, 1e-9), 4),
         "selection": ("dominant kernel by time: launches of the timed configuration summed per kernel name (as a rocprofv3 --stats row), "
                       "conv kernels included, no weighting; reported through that kernel's heaviest launch shape.  cu_share = "
                       "min(workgroups, 256) / 256 and frac_of_occupied_cus = frac / cu_share are extra fields")}
    e["workgroups"] = row["workgroups"]
    e["median_launch_us"] = row.get("median_launch_us")
    e["outlier_launches_dropped"] = row.get("outlier_launches_dropped")
    e["timer"] = ("live HIP-event pair per launch on the launch stream (fvit_prof_*); average over the launches within "
                  f"{OUTLIER} x the median of this (kernel, shape)")
    rp = rocprof_row(row["kernel"], row["workgroups"]) if pmc_file == PMC_FILE else None
    if rp is not None:
        e["avg_launch_us_rocprof"] = rp["avg_us"]
        # MFLOP / us = TFLOP/s; MB / us * 1e3 = GB/s
        rate = row["algorithmic_mflop_per_launch"] / rp["avg_us"] if row["bound"] == "mfma" else row["algorithmic_mbyte_per_launch"] / rp["avg_us"] * 1e3
        e["frac_rocprof"] = round(rate / e["peak"], 4)
        e["rocprof_source"] = f"{rp['file']}: {rp['name']} x {row['workgroups']} workgroups, {rp['calls']} calls"
        # r06: `frac` / `achieved` are what the TIMED hipGraph achieves -- the kernel-trace duration of the overlapped replays, where
        # the two stream shards' kernels stretch each other --, the serialized event-timer figure stays beside it as frac_serialized
        e["frac_serialized"], e["achieved_serialized"] = e["frac"], e["achieved"]
        e["frac"], e["achieved"] = e["frac_rocprof"], round(rate, 2)
        e["frac_source"] = "committed rocprofv3 kernel trace of the overlapped graph replays (avg_launch_us_rocprof); frac_serialized = live event timer of a serialized pass"
        # the committed kernel trace is of the overlapped graph replays (shards stretch each other), the event pair of a serialized pass: they differ
        # by ~10-15 % on short kernels.  More than EVENT_VS_ROCPROF_TOL apart means one of the two is not measuring this kernel: say so in the line.
        dev = abs(row["avg_launch_us"] - rp["avg_us"]) / max(rp["avg_us"], 1e-9)
        e["event_vs_rocprof"] = round(dev, 3)
        if dev > EVENT_VS_ROCPROF_TOL:
            e["timer_mismatch"] = (f"live event timer {row['avg_launch_us']} us vs committed rocprofv3 trace {rp['avg_us']} us differ by {dev * 100:.0f} % "
                                   f"(> {EVENT_VS_ROCPROF_TOL * 100:.0f} %): trust frac_rocprof, re-profile")
    pm = pmc_row(row["kernel"], row["workgroups"], pmc_file)
    if pm is not None:
        e["traffic"] = int(pm["hbm_traffic_mb"] * 1e6)
        e["traffic_over_algorithmic"] = round(pm["hbm_traffic_mb"] / max(row["algorithmic_mbyte_per_launch"], 1e-9), 2)
        e["traffic_source"] = (f"{pmc_file}: {pm['kernel']} x {pm['workgroups']} workgroups, read {pm['hbm_read_mb']} MB (2 x FETCH_SIZE) + "
                               f"write {pm['hbm_write_mb']} MB per launch, {pm.get('avg_us_under_pmc', '?')} us in that pass")
    return e


# ------------------------------------------------------------------------------------------------------------------------
# parity and CPU baseline (the oracle is imported ONLY here, after the timed region)
# ------------------------------------------------------------------------------------------------------------------------
def oracle_arch(model_name, model_kwargs):
    from tests.cases import CASES
    for c in CASES.values():
        if c["entry"] == model_name and c["kwargs"] == (model_kwargs or {}) and c["family"] == "init":
            return dict(c["arch"])
    return None


def parity_vs_oracle(cfg, logits_gpu, arch, indices, what, threads=None, chunk=32):
    """max-abs logits error over `indices` (r06: ALL images of the timed batch on one GPU) and the image that carries it; the oracle runs in chunks."""
    from oracle.model_reference import model_forward
    torch.set_num_threads(threads or min(32, os.cpu_count() or 1))
    t0 = time.perf_counter()
    ref = torch.cat([model_forward(cfg.sd_cpu, cfg.x_cpu[indices[i:i + chunk]], arch) for i in range(0, len(indices), chunk)])
    per_image = (logits_gpu[indices] - ref).abs().amax(dim=1)
    err = per_image.max().item()
    return {"logits_max_abs_err": float(f"{err:.3e}"), "logits_abs_max": round(ref.abs().max().item(), 4), "images": len(indices),
            "worst_image": int(indices[int(per_image.argmax())]), "median_image_err": float(f"{per_image.median().item():.3e}"),
            "relative": float(f"{err / max(ref.abs().max().item(), 1e-30):.3e}"), "oracle_s": round(time.perf_counter() - t0, 1), "vs": what}, ref


def cpu_baseline(cfg, arch, seconds):
    from oracle.model_reference import model_forward
    ncpu = os.cpu_count() or 1
    model = "?"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    out = {}
    gen = torch.Generator().manual_seed(77)
    for nb in (8, 64):
        xs = cfg.x_cpu[:nb] if nb <= cfg.batch else torch.randn(nb, 3, cfg.H, cfg.W, generator=gen)
        cands = sorted({t for t in (8, 16, 32, 64) if t <= ncpu} or {ncpu})
        best_t, best_dt = cands[0], float("inf")
        torch.set_num_threads(cands[0])
        model_forward(cfg.sd_cpu, xs[:8], arch)   # warm-up
        for t in cands:   # small-batch fp32 inference does not scale to hundreds of threads: pick the fastest of a few counts
            torch.set_num_threads(t)
            t1 = time.perf_counter()
            model_forward(cfg.sd_cpu, xs, arch)
            dt = time.perf_counter() - t1
            if dt < best_dt:
                best_t, best_dt = t, dt
        torch.set_num_threads(best_t)
        n, t_cpu = 0, 0.0
        while t_cpu < seconds and n < 50:
            t1 = time.perf_counter()
            model_forward(cfg.sd_cpu, xs, arch)
            t_cpu += time.perf_counter() - t1
            n += 1
        out[nb] = {"value": round(nb * n / t_cpu, 2), "unit": "images/s", "cores": best_t, "kind": "port", "cpu_model": model, "host_threads": ncpu,
                   "sample": f"{cfg.name} fp32 oracle (port of the reference CPU path), batch {nb}, {n} iterations ({t_cpu:.1f} s), "
                             f"{best_t} of {ncpu} host threads (fastest of {cands})"}
    head = dict(out[8])
    head["batch_64"] = out[64]
    return head


def run_secondary(args, dev):
    """BASELINE configs 3 and 5 (N = 1 only), each in TWO configurations, both timed and both checked on the same 8 images:
       value / parity   the PRECISE deploy plan (r05): two-term conv streams + two-term conv weights (DeployPlan.precise) + HAT operands f16x3 (two-term
                        weights AND activations, dual-tile GEMMs), stream shards, hipGraph -- the configuration that meets north_star's ABSOLUTE bar
                        (logits max-abs < 1e-3) on these models, whose logits reach |7| with the gamma ~ U(0.5, 1.5) test weights.  THE timed entry.
       fast             the 16-bit deploy plan with the headline's operand mode (1x MFMA work; a RELATIVE 1e-3 claim only), for comparison."""
    import copy
    from fastervit_amd import dp
    res = []
    # (name, batch, image size, model kwargs, steps in flight of the 16-bit plan, steps in flight of the precise plan): r06, whole-batch launches (one stream shard)
    # with 2 / 3 steps in flight -- call 12: FasterViT-4 3.34k -> 3.39k precise, 6.80k -> 7.02k 16-bit; any-res 275 -> 291 precise, 614 -> 657 16-bit images/s against
    # r05's 2 / 3 stream shards inside one graph (--secondary-streams N > 0 restores that form with --inflight 1)
    specs = [("faster_vit_4_224", 128, None, {}, 2, 2),
             ("faster_vit_4_any_res", 8, (576, 960), dict(resolution=[576, 960], window_size=[7, 7, 12, 6], ct_size=2), 3, 3)]
    for name, batch, hw, kw, inflight_fast, inflight_precise in specs:
        nstreams = nstreams_fast = args.secondary_streams or 1
        t0 = time.perf_counter()
        try:
            a1 = copy.copy(args)
            a1.mode, a1.conv_dtype, a1.operand, a1.no_graph, a1.precise = "deploy", "f16", "f16x3", False, True
            a1.inflight = inflight_precise if (args.inflight > 1 and not args.secondary_streams) else 1
            cfg = Config(a1, dev, 0, name, batch, hw=hw, model_kwargs=kw, streams=nstreams)
            cfg.prepare()
            elapsed = dp.timed_steps(cfg.step, args.secondary_steps, 3, torch.cuda.synchronize, None, dev)
            logits = cfg.logits()
            arch = oracle_arch(name, kw)
            par = refp = None
            idx = list(range(batch))   # r06: EVERY image of the timed batch (r05: 8)
            if arch is not None:
                par, refp = parity_vs_oracle(cfg, logits, arch, idx, f"CPU oracle fp32, all {batch} images of the timed batch", chunk=8 if hw else 16)
                par["meets_1e-3"] = bool(par["logits_max_abs_err"] < 1e-3)
                par["tolerance"] = "north_star: logits max-abs < 1e-3 (absolute)"
            shapes = profile_shapes(cfg, 1) if args.prof_steps > 0 else []
            sec_roof = None
            if shapes:
                dom, fam_ms = dominant_by_time(shapes_cu(shapes))
                sec_roof = roofline_entry(dom, "f16x3", fam_ms, pmc_file_for(name))
                if sec_roof is not None:
                    # Linear layers: three MFMA passes per algorithmic FLOP (x3); convs: two-term weights (2), the three downsamples 3 -> 2 as the floor
                    sec_roof["dominant_family"] = dominant_family(shapes, {"gemm": 3.0, "conv": 2.0})
            entry = {"workload": f"{name} inference, {cfg.H}x{cfg.W}, batch {batch}, synthetic weights (tests/synth.py init family)",
                     "config": "precise deploy plan: two-term conv streams and weights + HAT operands f16x3",
                     "value": round(batch * args.secondary_steps / elapsed, 1), "unit": "images/s", "steps": args.secondary_steps,
                     "ms_per_step": round(elapsed / args.secondary_steps * 1e3, 4), "dtype": "f16x3", "launch": cfg.launch_desc(), "parity": par,
                     "roofline": sec_roof, "roofline_shapes": shapes[:8]}
            del cfg
            torch.cuda.empty_cache()
            if arch is not None and not args.no_modes:
                try:
                    a2 = copy.copy(args)
                    a2.inflight = inflight_fast if (args.inflight > 1 and not args.secondary_streams) else 1
                    fc = Config(a2, dev, 0, name, batch, hw=hw, model_kwargs=kw, streams=nstreams_fast)   # same seeds: same weights and input
                    fc.prepare()
                    el2 = dp.timed_steps(fc.step, args.secondary_steps, 3, torch.cuda.synchronize, None, dev)
                    yp = fc.logits()
                    ep = (yp[idx] - refp).abs().max().item()   # the same (all) images as the precise plan
                    entry["fast"] = {"config": f"16-bit deploy plan, HAT operands {args.operand} (relative claim only)",
                                     "value": round(batch * args.secondary_steps / el2, 1), "unit": "images/s", "steps": args.secondary_steps,
                                     "ms_per_step": round(el2 / args.secondary_steps * 1e3, 3),
                                     "logits_max_abs_err": float(f"{ep:.3e}"), "logits_abs_max": round(refp.abs().max().item(), 4),
                                     "relative": float(f"{ep / max(refp.abs().max().item(), 1e-30):.3e}"), "meets_1e-3": bool(ep < 1e-3),
                                     "images": len(idx)}
                    del fc
                except Exception as e:
                    entry["fast"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            entry["wall_s"] = round(time.perf_counter() - t0, 1)
            res.append(entry)
        except Exception as e:  # a secondary config must never take the headline line down with it
            res.append({"workload": name, "error": f"{type(e).__name__}: {e}"[:300]})
        torch.cuda.empty_cache()
    return res


def run_train_step(args, dev, batch=64, steps=5, warmup=2):
    """SURVEY.md section 8 row f-4, the part one GPU can verify: one TRAINING step of faster_vit_0_224 = forward in train mode
    (BatchNorm batch statistics, stochastic depth; HAT stages = the unit-kernel chain as ONE autograd node each) + backward (kernel-sequence backward of
    fastervit_amd.hat_backward for the HAT stages, PyTorch-ROCm autograd for the conv side) + AdamW on ALL parameters, batch 64, fp32 master weights,
    16-bit MFMA operands.  What train.py:820-951 does per iteration minus data / scheduler / EMA plumbing (out of scope).  A secondary number: the
    training path is correct and trainable (tests/test_gpu_backward.py), not tuned."""
    import fastervit_amd
    import torch.nn.functional as F
    torch.manual_seed(0)
    model = fastervit_amd.create_model("faster_vit_0_224", drop_path_rate=0.1).to(dev).train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.05)
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(batch, 3, 224, 224, generator=g).to(dev)
    y = torch.randint(0, 1000, (batch,), generator=g).to(dev)
    losses = []

    def step():
        opt.zero_grad(set_to_none=True)
        loss = F.cross_entropy(model(x), y)
        loss.backward()
        opt.step()
        losses.append(loss.detach())

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ls = [round(float(v), 4) for v in losses]
    return {"workload": f"faster_vit_0_224 TRAINING step (forward train mode + backward + AdamW, all parameters), batch {batch}, 224x224, synthetic data",
            "value": round(batch * steps / el, 1), "unit": "images/s", "steps": steps, "warmup": warmup, "ms_per_step": round(el / steps * 1e3, 2),
            "loss_first_last": [ls[0], ls[-1]], "finite": bool(all(v == v and abs(v) < 1e9 for v in ls)),
            "note": "HAT stages: unit-kernel forward + kernel-sequence backward (hat_backward.py), conv side: PyTorch-ROCm autograd; not tuned"}


# ------------------------------------------------------------------------------------------------------------------------
# the printed line: ONE compact JSON object (driver-parsed; < 8 KB, tests/test_bench_launch.py); everything else goes to a file
# ------------------------------------------------------------------------------------------------------------------------
LINE_BUDGET = 8000
DETAIL_FILES = [os.path.join("gpurun_out", "bench_detail.json")]   # scratch (not tracked); --record PATH adds a copy meant to be committed
_ROOF_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "avg_launch_us", "median_launch_us",
              "outlier_launches_dropped", "avg_launch_us_rocprof", "frac_rocprof", "frac_serialized", "event_vs_rocprof", "timer_mismatch", "launches_per_step", "ms_per_step", "workgroups", "cu_share", "algorithmic_mflop_per_launch",
              "algorithmic_mbyte_per_launch", "timer")
_PAR_KEYS = ("logits_max_abs_err", "logits_abs_max", "relative", "images", "worst_image", "meets_1e-3", "images_per_s", "steps", "tolerance", "error", "per_rank", "runners_max_abs_diff")


def _pick(d, keys):
    return None if d is None else {k: d[k] for k in keys if k in d and d[k] is not None}


def compact_line(out, detail_path=None):
    """The driver-facing subset of the full record: the contract keys, the dominant-kernel roofline, the CPU baseline, parity of the timed
    mode and of the two-term bf16 mode, three numbers per secondary configuration, and where the full record went."""
    c = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "data")}
    cfg = out.get("config") or {}
    c["config"] = {k: cfg[k] for k in ("workload", "global_batch", "parallelism", "hat_operands", "launch", "streams", "join_from") if k in cfg}
    if "step_ms" in out:
        c["step_ms"] = _pick(out["step_ms"], ("n", "min", "median", "max"))
    c["roofline"] = _pick(out.get("roofline"), _ROOF_KEYS)
    cb = out.get("cpu_baseline")
    c["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "sample", "cpu_model"))
    if cb and cb.get("batch_64"):
        c["cpu_baseline"]["batch_64_images_per_s"] = cb["batch_64"]["value"]
    c["parity"] = _pick(out.get("parity"), _PAR_KEYS)
    for m in OPERAND_MODES + tuple(mm + "_down2" for mm in ALL_OPERAND_MODES) + ("bf16x3_precise",):
        if "parity_" + m in out:
            c["parity_" + m] = _pick(out["parity_" + m], _PAR_KEYS)
    for k in ("bf16_under_1e-3_images_per_s", "bf16_under_1e-3_with_margin_images_per_s", "hat_ms_per_step", "kernel_ms_per_step_serialized", "launches_per_step"):
        if k in out:
            c[k] = out[k]
    if out.get("secondary"):
        c["secondary"] = []
        for s in out["secondary"]:
            e = {"workload": s.get("workload", "?")[:80]}
            for k in ("value", "unit", "ms_per_step", "steps", "dtype", "error"):
                if k in s:
                    e[k] = s[k]
            if s.get("parity"):
                e["parity"] = _pick(s["parity"], _PAR_KEYS)
            if s.get("config"):
                e["config"] = s["config"][:90]
            if s.get("fast"):
                e["fast"] = _pick(s["fast"], ("value", "ms_per_step", "logits_max_abs_err", "relative", "meets_1e-3", "images", "error"))
            if s.get("precise"):   # records of r04 and earlier
                e["precise"] = _pick(s["precise"], ("value", "ms_per_step", "logits_max_abs_err", "meets_1e-3", "images", "error"))
            if s.get("roofline"):
                e["roofline"] = _pick(s["roofline"], ("kernel", "bound", "frac", "avg_launch_us", "traffic_over_algorithmic"))
                if s["roofline"].get("dominant_family"):
                    e["roofline"]["dominant_family"] = _pick(s["roofline"]["dominant_family"], ("family", "share_of_kernel_time", "frac", "mfma_issue_frac"))
                    e["roofline"]["dominant_family"]["family"] = e["roofline"]["dominant_family"]["family"].split(" ")[0]
            c["secondary"].append(e)
    if out.get("train_step"):
        c["train_step"] = _pick(out["train_step"], ("value", "unit", "ms_per_step", "steps", "loss_first_last", "finite", "error"))
        c["train_step"]["workload"] = "faster_vit_0_224 fwd+bwd+AdamW, batch 64"
    c["detail"] = detail_path
    line = json.dumps(c, separators=(",", ":"))
    if len(line) > LINE_BUDGET:   # never let prose grow the line past what the driver's stdout tail holds
        for k in ("train_step", "secondary", "step_ms", "parity_bf16x3_precise", "parity_f16x2", "parity_bf16", "hat_ms_per_step", "kernel_ms_per_step_serialized", "launches_per_step"):
            c.pop(k, None)
            line = json.dumps(c, separators=(",", ":"))
            if len(line) <= LINE_BUDGET:
                break
    return line


def emit(out, record=""):
    """Write the full record (gpurun_out/, plus --record PATH) and print the compact line as the LAST stdout line."""
    path = None
    for rel in DETAIL_FILES + ([record] if record else []):
        try:
            full = os.path.join(ROOT, rel)
            os.makedirs(os.path.dirname(full), exist_ok=True)
            with open(full, "w") as f:
                json.dump(out, f, indent=1)
            path = path or rel
        except OSError:
            pass
    sys.stdout.flush()
    print(compact_line(out, path), flush=True)


def launch_selftest(args, dp, rank, world):
    """The N-rank plumbing of this file without a GPU: gloo group, W + K trivial steps through dp.timed_steps, MAX / SUM
    reductions, ONE JSON line from rank 0 (tests/test_bench_launch.py)."""
    dist = dp.init_process_group("gloo")
    elapsed = dp.timed_steps(lambda: time.sleep(0.002), args.steps, args.warmup, lambda: None, dist, None)
    value = dp.whole_job_rate(args.batch * args.steps, elapsed, dist, None)
    ran = dp.ranks_that_ran(dist, None)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "launch selftest (no model, CPU, gloo)", "value": round(value, 1), "unit": "items/s", "n_gpus": ran,
                          "group_world_size": world, "steps": args.steps, "warmup": args.warmup}))


def main():
    args = parse()
    from fastervit_amd import dp
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as a plain `python bench.py --gpus N`: spawn the N ranks here (one process per GPU under torch.distributed.run)
        # instead of silently measuring one GPU
        if not args.launch_selftest and torch.cuda.device_count() < args.gpus:
            print(f"bench.py: --gpus {args.gpus} requested but this node exposes {torch.cuda.device_count()} HIP device(s); "
                  "refusing to run fewer ranks than asked for", file=sys.stderr)
            sys.exit(2)
        sys.exit(dp.launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    rank, local_rank, world = dp.env_world()
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or without a launcher)", file=sys.stderr)
        sys.exit(2)
    if args.launch_selftest:
        return launch_selftest(args, dp, rank, world)
    dist = dp.init_process_group("nccl")  # RCCL on ROCm; None when WORLD_SIZE == 1
    if dist is not None and dist.get_world_size() != args.gpus:
        print(f"bench.py: RCCL group has {dist.get_world_size()} ranks, expected {args.gpus}", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    mk = ast.literal_eval(args.model_kwargs) if args.model_kwargs else {}
    hw = tuple(int(v) for v in args.input_size.lower().split("x")) if args.input_size else None
    cfg = Config(args, dev, rank, args.model, args.batch, hw=hw, model_kwargs=mk, streams=args.streams)
    cfg.prepare()

    # W untimed + exactly K timed steps, barrier + synchronize on both sides, MAX over ranks
    elapsed = dp.timed_steps(cfg.step, args.steps, args.warmup, torch.cuda.synchronize, dist, dev)
    value = dp.whole_job_rate(args.batch * args.steps, elapsed, dist, dev)
    n_ran = dp.ranks_that_ran(dist, dev)   # ranks that actually ran the timed region (== RCCL's group size)
    logits_gpu = cfg.logits()
    runners_diff = getattr(cfg, "runners_max_abs_diff", None)   # (of the timed operand mode: the later legs re-use the attribute)

    def finish():
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()

    # N > 1 (r05): EVERY rank checks 8 of its own images (both ends of its batch = both stream shards) against the CPU oracle,
    # host threads capped to its share of the cores; rank 0 reports the per-rank errors and their maximum.  After the timed region, so the
    # oracle's CPU work cannot disturb it.
    rank_parity = None
    if dist is not None and not args.no_rank_parity and args.prof_steps > 0:
        arch_r = oracle_arch(args.model, mk)
        err_r = -1.0
        if arch_r is not None:
            idx_r = sorted(set(list(range(min(4, args.batch))) + list(range(max(args.batch - 4, 0), args.batch))))
            par_r, _ = parity_vs_oracle(cfg, logits_gpu, arch_r, idx_r, "", threads=max(1, min(16, (os.cpu_count() or 1) // world)))
            err_r = par_r["logits_max_abs_err"]
        per_rank = [float(f"{e:.3e}") for e in dp.gather_values(err_r, dist, dev)]
        if min(per_rank) >= 0:
            rank_parity = {"logits_max_abs_err": max(per_rank), "per_rank": per_rank, "images": 8, "meets_1e-3": bool(max(per_rank) < 1e-3),
                           "vs": "CPU oracle fp32, 8 images (first and last 4) of EVERY rank's batch, each rank on its own host threads"}

    if rank != 0:
        return finish()
    ms_per_step = elapsed / args.steps * 1e3
    H, W = cfg.H, cfg.W
    headline = (args.model, args.batch, H, W) == ("faster_vit_0_224", 256, 224, 224)
    out = {
        "metric": ("images/sec FasterViT-0 224x224 inference, bs=256/GPU" if headline
                   else f"images/sec {args.model} {H}x{W} inference, bs={args.batch}/GPU (secondary config)"),
        "value": round(value, 1), "unit": "images/s", "n_gpus": n_ran, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.operand,
        "data": "synthetic",
        "config": {"workload": f"{args.model} inference, {H}x{W}, batch {args.batch}/GPU, synthetic weights (tests/synth.py init family)",
                   "global_batch": args.batch * world, "parallelism": f"dp{world} (independent shards, no data-path collective)",
                   "rccl_world_size": (dist.get_world_size() if dist is not None else 1),
                   "hat_operands": args.operand,
                   "conv_side": (f"deploy plan: BN folded, {args.conv_dtype} channels_last, HIP conv3x3 (halo-tiled / row-band / implicit-GEMM) + fused stem + LayerNorm2d kernels"
                                 if cfg.deploy else (f"model(x) under autocast {args.conv_dtype}: automatic deploy plan" if args.mode == "auto"
                                                     else f"PyTorch-ROCm nn.Module forward, channels_last, autocast {args.conv_dtype}")),
                   "launch": cfg.launch_desc(), "streams": cfg.streams,
                   "join_from": (getattr(cfg.plan, "join_from", None) if cfg.plan is not None else None)},
    }
    if args.prof_steps <= 0:   # timeline runs under rocprofv3 (scripts/gpu_trace.sh): nothing but the timed region
        out["roofline"] = None
        print(compact_line(out))
        return finish()

    out["step_ms"] = step_dispersion(cfg, min(max(args.steps, 5), 30))
    shapes = profile_shapes(cfg, args.prof_steps)
    hat = [r for r in shapes if r["kind"] not in ("other", "conv3x3")]
    for r in shapes:
        r["cu_share"] = round(min(r["workgroups"], N_CU) / N_CU, 3)
        r["gpu_ms_per_step"] = round(r["ms_per_step"] * r["cu_share"], 4)
    dom, fam_ms = dominant_by_time(shapes)
    out["roofline"] = roofline_entry(dom, args.operand, fam_ms)
    out["roofline_shapes"] = shapes[:24]
    out["hat_ms_per_step"] = round(sum(r["ms_per_step"] for r in hat), 4)
    out["kernel_ms_per_step_serialized"] = round(sum(r["ms_per_step"] for r in shapes), 4)
    out["launches_per_step"] = sum(r["launches_per_step"] for r in shapes)
    if cfg.plan is not None and cfg.plan.streams > 1:
        # the same kernels with the GPU to themselves: one stream, whole-batch launches (kernel quality, not job throughput)
        n = cfg.plan.streams
        cfg.plan.streams = 1
        iso = profile_shapes(cfg, 1, serialize=False)
        cfg.plan.streams = n
        out["roofline_isolated"] = {"launch": "eager, 1 stream, whole-batch launches", "shapes": iso[:12]}

    cpu = parity = None
    arch = oracle_arch(args.model, mk)
    if world == 1 and arch is not None:   # N = 1 only: rank 0's host cores are shared with the other ranks otherwise
        sizes = [p.shape[0] for p in cfg.x_cpu.chunk(cfg.streams)] if cfg.streams > 1 else [args.batch]
        if cfg.plan is not None and getattr(cfg.plan, "shard_sizes", None) and sum(cfg.plan.shard_sizes) == args.batch:
            sizes = list(cfg.plan.shard_sizes)
        starts = [sum(sizes[:i]) for i in range(len(sizes))]
        idx = list(range(args.batch))   # r06: EVERY image the timed region processed (r05: 8 per stream shard)
        parity, ref_all = parity_vs_oracle(cfg, logits_gpu, arch, idx, f"CPU oracle fp32; all {args.batch} images of the timed batch ({len(sizes)} stream shards, starts {starts})")
        parity["weights"] = f"tests/synth.py family 'init', seed {WEIGHT_SEED} (reference init + gamma ~ U(0.5,1.5), BN statistics, biases)"
        parity["tolerance"] = "north_star: logits max-abs < 1e-3"
        # every other operand mode on the same images: its own error AND its own images/s (the runner is re-captured per mode:
        # same deploy plan, stream shards and hipGraph as the timed configuration; a few timed replays each)
        for other in [m for m in OPERAND_MODES if m != args.operand and not args.no_modes]:
            try:
                cfg.model.set_hat_operand_dtype(other)
                cfg.recompile()
                # bf16x2 is BASELINE cfg 2's literal dtype under the bar: the FULL --steps / --warmup region (r06); the others a quarter of it
                full = other == "bf16x2"
                nrep, nwarm = (args.steps, args.warmup) if full else (max(5, args.steps // 4), 2)
                el = dp.timed_steps(cfg.step, nrep, nwarm, torch.cuda.synchronize, None, dev)
                y = cfg.logits()
                per_image = (y[idx] - ref_all).abs().amax(dim=1)
                err = per_image.max().item()
                out["parity_" + other] = {"logits_max_abs_err": float(f"{err:.3e}"), "logits_abs_max": round(ref_all.abs().max().item(), 4),
                                          "images": len(idx), "worst_image": int(per_image.argmax()), "meets_1e-3": bool(err < 1e-3),
                                          "images_per_s": round(args.batch * nrep / el, 1), "steps": nrep, "warmup": nwarm,
                                          "vs": f"CPU oracle fp32, all {len(idx)} images, HAT operands {other} (conv side {args.conv_dtype}), "
                                                "same plan / stream shards / hipGraph as the timed configuration"}
                if full and err < 1e-3:
                    out["bf16_under_1e-3_images_per_s"] = out["parity_" + other]["images_per_s"]
            except Exception as e:
                out["parity_" + other] = {"error": f"{type(e).__name__}: {e}"[:300]}
        cfg.model.set_hat_operand_dtype(args.operand)
        if cfg.runner is not None and cfg.plan is not None and not args.no_modes:
            # the accuracy option of the 16-bit plan: two-term weights in the three Downsample.reduction convs (same operand mode otherwise); r06: also under
            # bf16x2 over the FULL --steps region -- bf16 operands with margin under the bar (bf16x2 alone sits AT the bar once all 256 images are checked)
            for mode_d, full in ((args.operand, False), ("bf16x2", True)):
                key = f"parity_{mode_d}_down2"
                try:
                    cfg.model.set_hat_operand_dtype(mode_d)
                    cfg.set_plan(down_weight_terms=2, sig=None)
                    cfg.recompile()
                    nrep, nwarm = (args.steps, args.warmup) if full else (max(5, args.steps // 4), 2)
                    el = dp.timed_steps(cfg.step, nrep, nwarm, torch.cuda.synchronize, None, dev)
                    y = cfg.logits()
                    per_image = (y[idx] - ref_all).abs().amax(dim=1)
                    err = per_image.max().item()
                    out[key] = {"logits_max_abs_err": float(f"{err:.3e}"), "images": len(idx), "worst_image": int(per_image.argmax()), "meets_1e-3": bool(err < 1e-3),
                                "images_per_s": round(args.batch * nrep / el, 1), "steps": nrep,
                                "vs": f"CPU oracle fp32, all {len(idx)} images; HAT operands {mode_d}, Downsample.reduction convs with two-term weights"}
                    if full and err < 1e-3:
                        out["bf16_under_1e-3_with_margin_images_per_s"] = out[key]["images_per_s"]
                except Exception as e:
                    out[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
            cfg.model.set_hat_operand_dtype(args.operand)
            cfg.set_plan(down_weight_terms=1, sig=None)
        cfg.recompile()
        if cfg.runner is not None and cfg.plan is not None and not args.no_modes and headline:
            # north_star's "bf16", literally and with margin (r05): the PRECISE conv plan on bf16 planes + HAT operands bf16x3 -- bf16 MFMA operands everywhere, every
            # stream and weight as two bf16 terms -- in the timed launch structure (its own runner: the plan's plane dtype is fixed at compile time)
            try:
                cfg.model.set_hat_operand_dtype("bf16x3")
                jf = args.join_from if (args.join_from > 0 and cfg.streams > 1) else None
                rb = cfg.model.compile_inference(cfg.x, dtype=torch.bfloat16, streams=cfg.streams, graph=not args.no_graph, join_from=jf, precise=True)
                nrep = max(5, args.steps // 4)
                el = dp.timed_steps((lambda: rb.graph.replay()) if rb.graph is not None else (lambda: rb(cfg.x)), nrep, 2, torch.cuda.synchronize, None, dev)
                torch.cuda.synchronize()
                y = rb.static_y.float().cpu()
                err = (y[idx] - ref_all).abs().max().item()
                out["parity_bf16x3_precise"] = {"logits_max_abs_err": float(f"{err:.3e}"), "images": len(idx), "meets_1e-3": bool(err < 1e-3),
                                                "images_per_s": round(args.batch * nrep / el, 1),
                                                "vs": "CPU oracle fp32, the images of 'parity'; precise conv plan on bf16 planes + HAT operands bf16x3 (bf16 everywhere)"}
                del rb
            except Exception as e:
                out["parity_bf16x3_precise"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            cfg.model.set_hat_operand_dtype(args.operand)
            torch.cuda.empty_cache()
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(cfg, arch, args.cpu_seconds)
    out["cpu_baseline"], out["parity"] = cpu, (parity if parity is not None else rank_parity)
    if runners_diff is not None and out["parity"] is not None:
        out["parity"]["runners_max_abs_diff"] = runners_diff   # steps in flight: logits of the runners against runner 0 on the same input (0.0 = bitwise equal)
    if world == 1 and headline and not args.no_secondary:
        del cfg
        torch.cuda.empty_cache()
        out["secondary"] = run_secondary(args, dev)
        if not args.no_train_step:
            try:
                out["train_step"] = run_train_step(args, dev)
            except Exception as e:   # never take the headline line down
                out["train_step"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    emit(out, args.record)
    finish()


if __name__ == "__main__":
    main()
