"""The N>1 path of the data-parallel driver on CPU: world_size 2, gloo backend."""
import os
import time

import torch
import torch.multiprocessing as mp

from fastervit_amd import dp


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    dist = dp.init_process_group("gloo")
    assert dist is not None and dist.get_world_size() == world
    calls = []

    def step():
        calls.append(1)
        time.sleep(0.01 * (rank + 1))  # rank 1 is the slow shard

    elapsed = dp.timed_steps(step, steps=5, warmup=2, sync=lambda: None, dist=dist)
    rate = dp.whole_job_rate(items_this_rank=5 * (3 + rank), elapsed_max=elapsed, dist=dist)
    vals = dp.gather_values(1e-4 * (rank + 1), dist)   # per-rank parity errors as bench.py --gpus N gathers them
    assert vals == [1e-4 * (r + 1) for r in range(world)]
    out.put((rank, len(calls), elapsed, rate))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_timing_and_aggregation():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, c0, e0, r0), (_, c1, e1, r1) = res
    assert c0 == c1 == 7                       # warmup + steps, on every rank
    assert abs(e0 - e1) < 1e-9 and e0 >= 0.1   # both ranks see the MAX (the slow rank: 5 * 20 ms)
    assert abs(r0 - r1) < 1e-9
    assert abs(r0 - (15 + 20) / e0) < 1e-6     # SUM of items / MAX time


def test_shard_bounds_cover_everything():
    for n, w in [(256, 8), (10, 3), (7, 8)]:
        spans = [dp.shard_bounds(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_single_process_needs_no_group():
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    assert dp.init_process_group("gloo") is None
    e = dp.timed_steps(lambda: None, steps=3, warmup=1, sync=lambda: None)
    assert dp.whole_job_rate(30, e) == 30 / e
    assert dp.gather_values(0.5) == [0.5]
