"""bench.py started as a plain `python bench.py --gpus N` must spawn its N ranks itself."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e["OMP_NUM_THREADS"] = "1"
    return e


def test_plain_launch_spawns_two_gloo_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--launch-selftest"],
                       capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["group_world_size"] == 2


def test_more_gpus_than_devices_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a box with fewer than 2 HIP devices")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert r.returncode != 0
    assert "refusing to run fewer ranks" in r.stderr


def test_world_size_mismatch_fails_loudly():
    e = _env()
    e.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--launch-selftest"],
                       capture_output=True, text=True, timeout=300, env=e, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
