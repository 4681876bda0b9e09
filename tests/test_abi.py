"""C-ABI surface and host logic that do not need a GPU."""
import ctypes
import os
import sys
import re

import pytest
import torch

from fastervit_amd import _lib, hat_runtime
from oracle import hat_reference as hr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.isfile(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def test_header_symbols_are_exported(lib):
    """Every function include/fvit_hip.h declares is exported by libfvit_hip.so (and vice versa for the binding); the fvit_debug_* entry points of
    its #ifdef FVIT_DIAG section are exported ONLY by the diagnosis build libfvit_hip_diag.so."""
    hdr = open(os.path.join(ROOT, "include", "fvit_hip.h")).read()
    a, b = hdr.index("#ifdef FVIT_DIAG"), hdr.index("#endif /* FVIT_DIAG */")
    diag_decl = set(re.findall(r"\b(fvit_debug_[a-z0-9_]+)\s*\(", hdr[a:b]))   # (the comments of that section mention product entry points)
    declared = set(re.findall(r"\b(fvit_[a-z0-9_]+)\s*\(", hdr[:a] + hdr[b:]))
    declared.discard("fvit_stream_t")
    assert declared == set(_lib.EXPORTED_SYMBOLS)
    assert diag_decl == set(_lib.DIAG_SYMBOLS) and all(s.startswith("fvit_debug_") for s in diag_decl)
    shipped = ctypes.CDLL(os.path.join(_lib.CSRC_DIR, "libfvit_hip.so"))
    for s in declared:
        assert hasattr(shipped, s), s
    for s in diag_decl:
        assert not hasattr(shipped, s), f"{s} is exported by the shipped library"
    diag = ctypes.CDLL(os.path.join(_lib.CSRC_DIR, "libfvit_hip_diag.so"))
    for s in declared | diag_decl:
        assert hasattr(diag, s), s


def test_shipped_library_has_no_ablation_knobs():
    """"ablate_skip" / "*_ablate" make kernels skip work (wrong results): the strings must not even be present in the shipped library."""
    blob = open(os.path.join(_lib.CSRC_DIR, "libfvit_hip.so"), "rb").read()
    assert b"ablate_skip" not in blob and b"conv_ablate" not in blob and b"mlp_ablate" not in blob
    assert b"ablate_skip" in open(os.path.join(_lib.CSRC_DIR, "libfvit_hip_diag.so"), "rb").read()


def test_abi_version_and_struct_sizes(lib):
    assert lib.fvit_abi_version() == _lib.FVIT_ABI_VERSION
    assert ctypes.sizeof(_lib.FvitStageDesc) == 20 * 4
    assert ctypes.sizeof(_lib.FvitAttnWeights) == 12 * 8 + 2 * 4
    assert ctypes.sizeof(_lib.FvitMlpWeights) == 9 * 8
    assert ctypes.sizeof(_lib.FvitBlockWeights) == 2 * 104 + 2 * 72 + 16 + 8
    assert ctypes.sizeof(_lib.FvitMapView) == 48


def test_attention_spad(lib):
    for S, want in [(16, 16), (36, 48), (49, 64), (53, 64), (60, 64), (148, 160), (196, 208), (129, 160)]:
        assert lib.fvit_attention_spad(S) == want


def test_workspace_bytes_and_descriptor_validation(lib):
    d = _lib.FvitStageDesc(batch=256, C=256, heads=8, dpad=32, ws=7, H=14, W=14, Hp=14, Wp=14, cw=2, hier=1, square=1,
                           hidden=1024, depth=6, do_propagation=0, operand_dtype=_lib.FVIT_F16, spad=64, gpad=16, weight_terms=1)
    n = lib.fvit_stage_workspace_bytes(ctypes.byref(d))
    d.weight_terms = 2   # hi + lo weight terms: same workspace (activations are single-rounded in every mode)
    assert lib.fvit_stage_workspace_bytes(ctypes.byref(d)) == n
    d.weight_terms = 0
    assert lib.fvit_stage_workspace_bytes(ctypes.byref(d)) == 0 and b"weight_terms" in lib.fvit_last_error()
    d.weight_terms = 1
    rows = 256 * 4 * 53
    assert n >= rows * (256 * 4 + 256 * 2 + 768 * 2 + 256 * 2 + 1024 * 2)
    assert n < 2 * rows * (256 * 4 + 256 * 2 + 768 * 2 + 256 * 2 + 1024 * 2)
    d.spad = 48  # inconsistent with fvit_attention_spad
    assert lib.fvit_stage_workspace_bytes(ctypes.byref(d)) == 0
    assert b"spad" in lib.fvit_last_error()
    d.spad, d.dpad = 64, 16  # dpad must be 32 or 64
    assert lib.fvit_stage_workspace_bytes(ctypes.byref(d)) == 0


@pytest.mark.parametrize("sr0,sr1,ws,cw", [(2, 2, 7, 2), (3, 5, 12, 2), (2, 4, 3, 2), (4, 4, 5, 1)])
def test_index_tables_reproduce_reference_permutations(sr0, sr1, ws, cw):
    """build_tables vs the oracle's ct_dewindow / ct_window / cat / nearest-upsample on tagged tensors."""
    tb = hat_runtime.build_tables(sr0, sr1, ws, cw, True)
    nW, S, G, ncw = tb["nW"], tb["S"], tb["G"], tb["ncw"]
    # X rows tagged with their own index; carrier rows of X hold windowed carrier p at (p // ncw) * S + p % ncw
    X = torch.arange(nW * S, dtype=torch.float32).view(nW * S, 1)
    ct_windowed = torch.stack([X[(p // ncw) * S + p % ncw] for p in range(G)]).view(1, G, 1)
    raster = hr.ct_dewindow(ct_windowed, cw * sr0, cw * sr1, cw)
    assert torch.equal(X[tb["ct_src"].long()].view(1, G, 1), raster)
    # after the carrier branch: R (raster) -> ct_window -> cat in front of the local tokens
    R = torch.arange(1000, 1000 + G, dtype=torch.float32).view(1, G, 1)
    ctw = hr.ct_window(R, cw * sr0, cw * sr1, cw).reshape(nW, ncw, 1)
    xloc = X.view(nW, S, 1)[:, ncw:]
    cat = torch.cat((ctw, xloc), dim=1).reshape(nW * S)
    src = tb["ln1_src"].long()
    got = torch.where(src >= 0, X.view(-1)[src.clamp(min=0)], R.view(-1)[(-src - 1).clamp(min=0)])
    assert torch.equal(got, cat)
    add = tb["ln1_add"].view(nW, S)
    assert (add[:, :ncw] == -1).all() and torch.equal(add[:, ncw:], torch.arange(ws * ws).expand(nW, -1).int())
    # propagation: nearest upsample cw -> ws
    img = torch.arange(ncw, dtype=torch.float32).view(1, 1, cw, cw)
    up = torch.nn.functional.interpolate(img, size=(ws, ws), mode="nearest").reshape(-1)
    assert torch.equal(tb["up_idx"].float(), up)


def test_product_fails_loudly_without_gpu():
    """No CPU fallback: a transformer stage on a CPU tensor raises, it does not silently compute."""
    import fastervit_amd
    m = fastervit_amd.create_model("faster_vit_0_224", depths=[1, 1, 1, 1], dim=16, in_dim=16, num_heads=[1, 1, 2, 4]).eval()
    with torch.no_grad(), pytest.raises(RuntimeError, match="HIP device"):
        m(torch.randn(1, 3, 224, 224))
    blk = m.levels[3].blocks[0]
    with torch.no_grad(), pytest.raises(RuntimeError, match="HIP device"):
        blk(torch.randn(1, 49, 128), None)


def test_product_does_not_import_oracle():
    """The shipped package never references oracle/ (it is test infrastructure)."""
    pkg = os.path.join(ROOT, "fastervit_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f


def test_kernels_have_no_lds_pipeline_lane_exchange():
    """r02 (profiles/r02_repeatability_hunt.log): a __shfl_xor (ds_bpermute_b32, an LDS-pipeline instruction) issued while LDS-DMA of
    sibling waves was landing returned wrong lane values when kernels of several streams shared a CU.  Lane reductions go through the
    VALU helpers of fvit_common.h (v_permlane16/32_swap + DPP); only the diagnosis row-hash kernel, which has no LDS-DMA beside it,
    still shuffles."""
    csrc = os.path.join(ROOT, "fastervit_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".h")):
            continue
        for ln in open(os.path.join(csrc, f)).read().splitlines():
            code = ln.split("//")[0]
            if re.search(r"__shfl|ds_bpermute|ds_permute|ds_swizzle", code):
                assert f == "fvit_glue.hip" and "h ^=" in code, f"{f}: {ln.strip()}"


def test_registry_api():
    import fastervit_amd
    from fastervit_amd.models import registry
    assert len(fastervit_amd.list_models()) == 22
    assert fastervit_amd.is_model("faster_vit_4_21k_384_any_res")
    assert registry.list_models("faster_vit_[0-6]_224") == [f"faster_vit_{i}_224" for i in range(7)]
    assert registry.is_model_pretrained("faster_vit_0_224")
    m = fastervit_amd.create_model("faster_vit_0_224", depths=[1, 1, 1, 1], dim=16, in_dim=16, num_heads=[1, 1, 2, 4],
                                   num_classes=10)
    assert m.num_classes == 10 and m.head.out_features == 10
    assert m.pretrained_cfg["input_size"] == (3, 224, 224) and m.default_cfg is m.pretrained_cfg
    assert m.no_weight_decay_keywords() == {"rpb"}


def test_checkpoint_roundtrip(tmp_path):
    import fastervit_amd
    kw = dict(depths=[1, 1, 1, 1], dim=16, in_dim=16, num_heads=[1, 1, 2, 4])
    m = fastervit_amd.create_model("faster_vit_0_224", **kw)
    path = str(tmp_path / "ck.pth.tar")
    torch.save({"state_dict": {"module." + k: v for k, v in m.state_dict().items()}}, path)
    m2 = fastervit_amd.create_model("faster_vit_0_224", checkpoint_path=path, **kw)
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    m3 = fastervit_amd.create_model("faster_vit_0_224", **kw)
    m3._load_state_dict(path, strict=True)
    assert torch.equal(m3.head.weight, m.head.weight)


def test_committed_pmc_traffic_file_covers_the_default_kernels():
    """bench.py fills roofline.traffic from the committed PMC passes (it cannot sample counters on itself); the file must have been
    produced by the CURRENT default path: a row for every HAT kernel family the FasterViT-0 deploy plan launches (r01: it went stale
    silently after a kernel was replaced)."""
    import json
    import bench
    path = os.path.join(ROOT, bench.PMC_FILE)
    assert os.path.exists(path), path
    rows = json.load(open(path))["kernels"]
    fams = {r["kernel"].split("<")[0] for r in rows}
    for fam in ("winmlp_kernel", "attnblk_kernel", "ctblk8_kernel", "winblk_kernel", "conv3x3_kernel", "conv3x3_c64_halo_kernel", "conv3x3_c128_band_kernel",
                "stem_fused_kernel"):
        assert fam in fams, f"{bench.PMC_FILE} has no row of {fam}: re-run scripts/gpu_r2_evidence.sh and copy the new file"
    dom = [r for r in rows if r["kernel"].startswith("winmlp_kernel<f16,256") and r["workgroups"] == 848]   # whole-batch launches (r06: two steps in flight; r04 / r05: 424 = 128-image shards)
    assert dom and dom[0]["hbm_traffic_mb"] > 0


def test_inference_runners_have_no_cpu_path_and_bench_times_steps_in_flight():
    """compile_inference / pipelined_inference refuse a CPU example (no fallback); bench.py's default launch structure is the r06 one: two whole-batch steps in
    flight (--inflight 2, one stream shard, no join), the r05 structure stays reachable by flags."""
    import fastervit_amd
    import bench
    m = fastervit_amd.create_model("faster_vit_0_224").eval()
    x = torch.zeros(2, 3, 224, 224)
    with pytest.raises(RuntimeError, match="HIP device"):
        m.compile_inference(x)
    with pytest.raises(RuntimeError, match="HIP device"):
        m.pipelined_inference(x, depth=2)
    old = sys.argv
    try:
        sys.argv = ["bench.py"]
        a = bench.parse()
        assert (a.inflight, a.streams, a.join_from, a.batch, a.model) == (2, 1, 0, 256, "faster_vit_0_224")
        sys.argv = ["bench.py", "--inflight", "1", "--streams", "2", "--join-from", "3"]
        a = bench.parse()
        assert (a.inflight, a.streams, a.join_from) == (1, 2, 3)
    finally:
        sys.argv = old


def test_hat_backward_has_no_cpu_path():
    """fastervit_amd.hat_backward (sub-block backward on the HIP kernels) raises on CPU tensors instead of falling back."""
    from fastervit_amd import hat_backward
    x = torch.zeros(4, 256)
    with pytest.raises(RuntimeError, match="HIP device"):
        hat_backward.mlp_block_backward(x, x, x[0], x[0], torch.zeros(1024, 256), torch.zeros(1024), torch.zeros(256, 1024), torch.zeros(256), None, None)
    with pytest.raises(RuntimeError, match="HIP device"):
        hat_backward.attn_block_backward(x, x, x[0], x[0], torch.zeros(768, 256), None, torch.zeros(256, 256), torch.zeros(256), None, None, 8, 4, None)


def test_enable_hat_backward_flags_and_cpu_behaviour():
    """model.enable_hat_backward() marks the transformer levels only; on CPU tensors the stages still raise (no CPU path), and switching it off restores
    the forward-only contract."""
    import fastervit_amd
    model = fastervit_amd.create_model("faster_vit_0_224").eval()
    assert not any(lvl.__dict__.get("hat_backward", False) for lvl in model.levels)
    assert model.enable_hat_backward(True) is model
    assert [bool(lvl.__dict__.get("hat_backward", False)) for lvl in model.levels] == [False, False, True, True]
    x = torch.randn(1, 3, 224, 224, requires_grad=True)
    with pytest.raises(RuntimeError):
        model(x)                       # HAT stages need the HIP device
    model.enable_hat_backward(False)
    assert not any(lvl.__dict__.get("hat_backward", False) for lvl in model.levels)
