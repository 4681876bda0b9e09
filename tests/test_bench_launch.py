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


def test_roofline_selection_is_dominant_kernel_by_time():
    """`roofline` = the kernel NAME with the largest summed time (conv kernels included, no CU weighting), reported through
    its heaviest launch shape."""
    sys.path.insert(0, ROOT)
    import bench

    def row(kernel, kind, wg, ms, n=1, mflop=1.0):
        return dict(kernel=kernel, kind=kind, workgroups=wg, launches_per_step=n, avg_launch_us=ms * 1e3 / n, ms_per_step=ms,
                    algorithmic_mflop_per_launch=mflop, algorithmic_mbyte_per_launch=1.0, flop_per_byte=1.0, bound="mfma", tflops=1.0, gbs=1.0, frac=0.1)
    shapes = [row("winmlp_kernel<256>", "mlp_fused", 285, 0.60), row("winmlp_kernel<512>", "mlp_fused", 66, 0.45), row("winmlp_kernel<512>", "mlp_fused", 65, 0.40),
              row("conv3x3_kernel<2,2,4>", "conv3x3", 527, 0.50), row("other", "other", 0, 9.0, mflop=0.0)]
    dom, fam_ms = bench.dominant_by_time(bench.shapes_cu(shapes))
    assert dom["kernel"] == "winmlp_kernel<512>" and dom["workgroups"] == 66 and abs(fam_ms - 0.85) < 1e-9
    shapes[3]["ms_per_step"] = 0.9    # a conv kernel can be the dominant one: no exclusion
    dom, _ = bench.dominant_by_time(shapes)
    assert dom["kernel"] == "conv3x3_kernel<2,2,4>"
    e = bench.roofline_entry(dom, "f16", 0.9)
    assert e["cu_share"] == 1.0 and "dominant kernel by time" in e["selection"]


def test_printed_line_is_compact_and_carries_the_contract():
    """The driver parses the LAST stdout line of bench.py; r03's 26 KB line was truncated by its stdout tail (BENCH_r03 parsed = null).
    The full r03 record goes through the compaction the live run uses: < 8 KB, valid JSON, contract keys + roofline + cpu_baseline."""
    import bench
    full = json.loads(open(os.path.join(ROOT, "profiles", "r03_bench_final.json")).read().strip().splitlines()[-1])
    line = bench.compact_line(full, "gpurun_out/bench_detail.json")
    assert len(line) < bench.LINE_BUDGET <= 8000 and "\n" not in line
    c = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "parity"):
        assert k in c, k
    assert c["value"] == full["value"] and "workload" in c["config"] and "model" not in c["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in c["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c["cpu_baseline"], k
    assert len(c["secondary"]) == 2 and all("value" in s and "parity" in s for s in c["secondary"])
    # a record bloated with prose still fits: optional blocks are dropped before the contract keys
    full["secondary"] = full["secondary"] * 20
    assert len(bench.compact_line(full)) < bench.LINE_BUDGET


def test_one_slow_launch_does_not_flip_the_dominant_kernel():
    """in the r04 driver run ONE outlier launch moved ctblk8's mean event time from 34 to 58 us and the parsed roofline named
    the wrong kernel.  The per-launch duration is now the average of the launches within 1.5 x the median of their (kernel, shape)."""
    import bench
    recs = []
    for step in range(3):
        for i in range(12):
            recs.append(dict(kind="win_mlp", name="winmlp_kernel<256>", grid=424, flops=28.454e9, bytes=56.6e6, ms=0.0545))
            ms = 0.900 if (step == 0 and i == 0) else 0.0326      # one launch stalls for 0.9 ms
            recs.append(dict(kind="ct_block", name="ctblk8_kernel<256,G16>", grid=128, flops=0.9e9, bytes=5.8e6, ms=ms))
    shapes = bench.summarize_launches(recs, 3)
    dom, fam_ms = bench.dominant_by_time(bench.shapes_cu(shapes))
    assert dom["kernel"] == "winmlp_kernel<256>" and dom["launches_per_step"] == 12
    ct = next(r for r in shapes if r["kernel"].startswith("ctblk8"))
    assert abs(ct["avg_launch_us"] - 32.6) < 0.1 and ct["outlier_launches_dropped"] == 1
    # ... and a live timer that disagrees with the committed kernel trace by more than 25 % is flagged in the line
    e = bench.roofline_entry(dict(dom, avg_launch_us=100.0), "f16", fam_ms)
    if "avg_launch_us_rocprof" in e:
        assert "timer_mismatch" in e and e["event_vs_rocprof"] > 0.25


def test_r05_record_compacts_with_the_new_entries():
    """The committed r05 full record through the live compaction: the secondary entries carry the TIMED precise plan (`parity` absolute, meets_1e-3) with the
    16-bit plan beside it (`fast`), the bf16-everywhere leg and the training-step entry are in the line, and it still fits the driver's tail."""
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_final_detail.json")))
    line = bench.compact_line(full, "gpurun_out/bench_detail.json")
    assert len(line) < bench.LINE_BUDGET and "\n" not in line
    c = json.loads(line)
    assert c["config"]["streams"] == 2 and c["config"]["join_from"] == 3
    assert c["roofline"]["kernel"].startswith("winmlp_kernel<256>") and c["roofline"]["outlier_launches_dropped"] == 0 and "timer_mismatch" not in c["roofline"]
    assert c["parity"]["logits_max_abs_err"] < 1e-3
    assert c["parity_bf16x3_precise"]["meets_1e-3"] and c["parity_bf16x3_precise"]["logits_max_abs_err"] < 7.5e-4
    assert len(c["secondary"]) == 2
    for s in c["secondary"]:
        assert s["dtype"] == "f16x3" and s["parity"]["meets_1e-3"] and s["parity"]["logits_max_abs_err"] < 7.5e-4   # absolute, >= 25 % margin
        assert s["fast"]["value"] > s["value"] and not s["fast"]["meets_1e-3"]
    assert c["train_step"]["finite"] and c["train_step"]["value"] > 0
    assert json.loads(open(os.path.join(ROOT, "profiles", "r05_bench_final.json")).read().strip())["value"] == c["value"]


def test_r06_final_record_carries_the_contract():
    """The committed r06 final record (profiles/r06_bench_final*.json, the driver-form `python bench.py` of the final evidence run) through the live compaction: the
    contract keys, parity on ALL timed images of every configuration, the roofline of the timed region consistent with the committed kernel trace, the CPU baseline,
    and the printed line == the compaction of the full record."""
    import csv
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_final_detail.json")))
    line = bench.compact_line(full, "gpurun_out/bench_detail.json")
    assert len(line) < bench.LINE_BUDGET and "\n" not in line
    c = json.loads(line)
    printed = json.loads(open(os.path.join(ROOT, "profiles", "r06_bench_final.json")).read().strip().splitlines()[-1])
    assert printed["value"] == c["value"] and printed["ms_per_step"] == c["ms_per_step"]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in c, k
    assert c["n_gpus"] == 1 and c["scaling"] == "weak" and c["vs_baseline"] is None and c["data"] == "synthetic" and c["higher_is_better"] is True
    assert abs(c["value"] - 256 * 1e3 / c["ms_per_step"]) < 0.01 * c["value"]                     # images/s = batch / step time
    assert "faster_vit_0_224" in c["config"]["workload"] and c["config"]["global_batch"] == 256
    assert c["parity"]["images"] == 256 and c["parity"]["logits_max_abs_err"] < 1e-3 and c["parity"]["runners_max_abs_diff"] == 0.0
    assert len(c["secondary"]) == 2
    assert c["secondary"][0]["parity"]["images"] == 128 and c["secondary"][1]["parity"]["images"] == 8
    for s in c["secondary"]:
        assert s["parity"]["meets_1e-3"] and s["parity"]["logits_max_abs_err"] < 5e-4 and s["fast"]["value"] > s["value"]
    r = c["roofline"]
    assert r["bound"] == "mfma" and r["peak"] == 2500.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"] > 0
    # the line's fraction is the kernel-trace (timed-region) figure: recompute it from the committed rocprofv3 summary of the same run
    rows = [x for x in csv.DictReader(open(os.path.join(ROOT, "profiles", "r06_bench_final_fvit_kernels_by_shape.csv")))
            if x["Name"].startswith("winmlp_kernel<f16,256,") and x["Workgroups"] == str(r["workgroups"])]
    assert len(rows) == 1
    tf = r["algorithmic_mflop_per_launch"] / float(rows[0]["AverageUs"])                         # MFLOP / us = TFLOP/s
    assert abs(tf / r["peak"] - r["frac"]) < 0.05 * r["frac"]
    assert c["cpu_baseline"]["kind"] == "port" and c["cpu_baseline"]["value"] > 0 and c["cpu_baseline"]["cores"] >= 1
