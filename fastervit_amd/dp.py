"""Data-parallel inference driver: one process per GPU, independent image shards, no collective on
the data path (the reference's nn.DataParallel scatter/gather, validate.py:243-244,
is replaced by per-process shards).  The only collectives are the start/stop barriers and the
reductions that turn per-rank timings into one whole-job figure; on ROCm backend "nccl" is RCCL
over xGMI, on CPU tests use "gloo".
"""
from __future__ import annotations

import os
import time
from typing import Callable, Optional, Tuple


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_process_group(backend: str = "nccl"):
    """Initialise torch.distributed from the environment when WORLD_SIZE > 1; returns the module or None."""
    _, _, world = env_world()
    if world <= 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, init_method="env://")
    return dist


def free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(script: str, argv, nproc: int, env: Optional[dict] = None) -> int:
    """Re-execute ``script argv`` as ``nproc`` ranks of ONE node under ``torch.distributed.run`` (rendezvous on 127.0.0.1, a free
    port), one process per GPU; returns the launcher's exit code.  Used by ``bench.py --gpus N`` when it is started as a plain
    ``python bench.py`` (no WORLD_SIZE in the environment): the ranks are spawned here instead of silently running one."""
    import subprocess
    import sys
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(nproc)}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), script] + list(argv)
    e = dict(os.environ)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver (RCCL / cross-process tensor sharing)
    e.setdefault("OMP_NUM_THREADS", "8")
    e.update(env or {})
    return subprocess.call(cmd, env=e)


def ranks_that_ran(dist=None, device=None) -> int:
    """Number of ranks that reached this point (SUM of ones over the process group; 1 without a group)."""
    if dist is None:
        return 1
    import torch
    t = torch.ones(1, dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(round(t.item()))


def shard_bounds(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced split of ``n_items`` (strong-scaling use: one global batch over N GPUs)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def timed_steps(step: Callable[[], object], steps: int, warmup: int, sync: Callable[[], None], dist=None,
                device=None) -> float:
    """Run ``warmup`` untimed then exactly ``steps`` timed steps bracketed by barrier + sync on both
    sides; returns the MAX elapsed seconds over ranks."""
    import torch
    for _ in range(warmup):
        step()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def whole_job_rate(items_this_rank: int, elapsed_max: float, dist=None, device=None) -> float:
    """Aggregate items/s over all ranks: SUM of the items every rank processed / MAX elapsed."""
    import torch
    total = float(items_this_rank)
    if dist is not None:
        t = torch.tensor([total], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total = float(t.item())
    return total / elapsed_max


def gather_values(value: float, dist=None, device=None) -> list:
    """One float per rank, in rank order, on every rank ([value] without a group): per-rank parity errors of ``bench.py --gpus N``."""
    if dist is None:
        return [float(value)]
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    outs = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return [float(o.item()) for o in outs]
