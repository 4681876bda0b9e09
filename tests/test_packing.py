"""CPU checks of the MFMA fragment-order weight packers (hat_runtime.frag_pack_*): every fused kernel (attention block, fused / N-split MLP,
carrier-branch and window kernels) reads these images linearly, so the documented element mapping of include/fvit_hip.h is pinned here
against a plain index formula on random weights."""
import pytest
import torch

from fastervit_amd import hat_runtime as hr


def kch(kk, g, e):
    return (kk >> 1) * 64 + g * 16 + (kk & 1) * 8 + e


@pytest.mark.parametrize("C", [256, 512])
def test_kslot_channels_is_a_permutation_with_the_documented_formula(C):
    k = hr.kslot_channels(C)
    assert sorted(k.tolist()) == list(range(C))
    for slot in (0, 7, 8, 31, 32, 33, 95, C - 1):
        kk, g, e = slot >> 5, (slot >> 3) & 3, slot & 7
        assert int(k[slot]) == kch(kk, g, e)


@pytest.mark.parametrize("C,hid", [(256, 1024), (512, 2048)])
def test_frag_pack_fc1_and_fc2(C, hid):
    g_ = torch.Generator().manual_seed(C)
    w1 = torch.randn(hid, C, generator=g_)
    w2 = torch.randn(C, hid, generator=g_)
    p1 = hr.frag_pack_fc1(w1).reshape(hid // 32, 2, C // 32, 64, 8)     # lane = 16 g + s
    p2 = hr.frag_pack_fc2(w2).reshape(hid // 32, C // 16, 64, 8)
    assert p1.shape == (hid // 32, 2, C // 32, 64, 8) and p2.shape == (hid // 32, C // 16, 64, 8)
    for (j, hb, kk, g, s, e) in [(0, 0, 0, 0, 0, 0), (3, 1, 5, 2, 9, 7), (hid // 32 - 1, 1, C // 32 - 1, 3, 15, 7), (7, 0, 1, 1, 4, 3)]:
        assert p1[j, hb, kk, 16 * g + s, e] == w1[j * 32 + hb * 16 + s, kch(kk, g, e)]
    for (j, cb, g, s, e) in [(0, 0, 0, 0, 0), (5, 9, 2, 11, 6), (hid // 32 - 1, C // 16 - 1, 3, 15, 7), (2, 3, 1, 7, 4)]:
        ch = (cb >> 2) * 64 + (s >> 2) * 16 + (cb & 3) * 4 + (s & 3)
        col = j * 32 + (e >> 2) * 16 + 4 * g + (e & 3)
        assert p2[j, cb, 16 * g + s, e] == w2[ch, col]
    # the packings are bijections of the weight elements
    assert torch.equal(p1.flatten().sort().values, w1.flatten().sort().values)
    assert torch.equal(p2.flatten().sort().values, w2.flatten().sort().values)


@pytest.mark.parametrize("C,heads", [(256, 8), (512, 16)])
def test_frag_pack_qkv(C, heads):
    g_ = torch.Generator().manual_seed(heads)
    w = torch.randn(3 * C, C, generator=g_)
    p = hr.frag_pack_qkv(w, heads).reshape(heads, 6, C // 32, 64, 8)
    assert p.shape == (heads, 6, C // 32, 64, 8)
    for (h, ub, kk, g, s, e) in [(0, 0, 0, 0, 0, 0), (heads - 1, 5, C // 32 - 1, 3, 15, 7), (3, 2, 4, 1, 6, 5), (5, 4, 7, 2, 13, 1)]:
        row = (ub >> 1) * C + h * 32 + (ub & 1) * 16 + s
        assert p[h, ub, kk, 16 * g + s, e] == w[row, kch(kk, g, e)]
    assert torch.equal(p.flatten().sort().values, w.flatten().sort().values)


def test_two_term_weight_packing_reconstructs_the_weights():
    """x2 operand modes (FvitStageDesc.weight_terms = 2): row-major arrays hold [hi | lo] along K, fragment-order arrays two images back to
    back; hi + lo reproduces the fp32 weight to ~2^-16 (bf16) / 2^-21 (fp16) relative, and the hi part IS the single-term packing."""
    import torch
    from fastervit_amd import hat_runtime
    g = torch.Generator().manual_seed(3)
    w = torch.randn(96, 64, generator=g) * 0.05
    for dt, rel in ((torch.bfloat16, 2.0 ** -15), (torch.float16, 2.0 ** -20)):
        k1, k2 = hat_runtime._Keep(dt, 1), hat_runtime._Keep(dt, 2)
        a1, a2 = k1.op16(w), k2.op16(w)
        assert a1.shape == (96, 64) and a2.shape == (96, 128) and a2.dtype == dt and a2.is_contiguous()
        assert torch.equal(a2[:, :64], a1)                                   # hi term = the single-term array
        rec = a2[:, :64].float() + a2[:, 64:].float()
        assert (rec - w).abs().max().item() <= rel * w.abs().max().item()
        assert (a1.float() - w).abs().max().item() > 4 * (rec - w).abs().max().item()
        f1, f2 = k1.frag16(w), k2.frag16(w)
        assert f1.numel() == w.numel() and f2.numel() == 2 * w.numel()
        assert torch.equal(f2[:w.numel()], f1)                               # [hi image | lo image]
        assert torch.equal(f2[w.numel():].float(), (w - f1.view_as(w).float()).to(dt).float().reshape(-1))


def test_operand_modes_table():
    from fastervit_amd import hat_runtime, _lib
    assert set(hat_runtime.OPERAND_MODES) == {"f16", "bf16", "f16x2", "bf16x2", "f16x3", "bf16x3"}
    assert hat_runtime._OP["f16x3"][2] == 3
    assert hat_runtime._OP["bf16x2"][0] == _lib.FVIT_BF16 and hat_runtime._OP["bf16x2"][2] == 2
    assert hat_runtime._OP["f16"][2] == 1


def test_conv128_fragment_stream_follows_its_documented_element_mapping():
    """frag_pack_conv128 (fvit_conv3x3_c128_band's w_frag, include/fvit_hip.h): element e of lane 16 g + s of fragment (wave, step, ni) is
    weight[32 wave + (s >> 2) * 8 + ni * 4 + (s & 3)][step * 32 + 8 g + e]; every weight element appears exactly once."""
    from fastervit_amd.conv_runtime import frag_pack_conv128
    w = torch.arange(128 * 1152, dtype=torch.float32).view(128, 1152)
    f = frag_pack_conv128(w)
    assert tuple(f.shape) == (4, 36, 2, 64, 8)
    gen = torch.Generator().manual_seed(3)
    for _ in range(200):
        wave, step, ni, lane, e = (int(torch.randint(0, n, (1,), generator=gen)) for n in (4, 36, 2, 64, 8))
        g, s = lane >> 4, lane & 15
        assert f[wave, step, ni, lane, e].item() == w[32 * wave + (s >> 2) * 8 + ni * 4 + (s & 3), step * 32 + 8 * g + e].item()
    assert torch.equal(torch.sort(f.reshape(-1)).values, w.reshape(-1))


def test_three_term_packing_and_the_gemm_column_map():
    """x3 modes (FvitStageDesc.weight_terms = 3): weight rows are packed [hi | lo | hi]; the GEMM reads activation column
    (k >= ka ? k - ka : k) of a [hi | lo] row, i.e. segments [hi | hi | lo] -- together hi.hi + a_hi.w_lo + a_lo.w_hi.  Fragment-order
    images (fused kernels) are not produced in this mode."""
    from fastervit_amd import hat_runtime
    g = torch.Generator().manual_seed(5)
    w = torch.randn(7, 64, generator=g) * 0.05
    a = torch.randn(3, 64, generator=g)
    keep = hat_runtime._Keep(torch.float16, 3)
    wp = keep.op16(w)
    assert tuple(wp.shape) == (7, 192) and keep.frag16(w) is None
    hi = w.half()
    lo = (w - hi.float()).half()
    assert torch.equal(wp[:, :64], hi) and torch.equal(wp[:, 64:128], lo) and torch.equal(wp[:, 128:], hi)
    ah = a.half()
    al = (a - ah.float()).half()
    arow = torch.cat([ah, al], dim=1).float()                      # the [hi | lo] activation row
    k = torch.arange(192)
    acol = torch.where(k >= 64, k - 64, k)                          # csrc/fvit_gemm.hip: acol
    y = arow[:, acol] @ wp.float().t()
    exact = a.double() @ w.double().t()
    single = ah.float() @ hi.float().t()
    assert (y.double() - exact).abs().max() < 2e-6 < (single.double() - exact).abs().max()


def test_x3_modes_are_refused_at_set_time_for_geometries_they_do_not_cover():
    """the two-term-activation modes cover head_dim <= 96 and no Dropout on the softmax probabilities (r06: any window length -- the long
    attention kernel has two-term instances); anything else raises in set_hat_operand_dtype, naming the level -- not at the first forward."""
    import pytest
    import fastervit_amd
    m = fastervit_amd.create_model("faster_vit_0_224")
    m.set_hat_operand_dtype("f16x3")            # 7x7 windows + 4 carriers, head_dim 32: covered
    assert m.hat_operand_dtype == "f16x3"
    big = fastervit_amd.create_model("faster_vit_0_any_res", resolution=[512, 512], window_size=[7, 7, 16, 8], ct_size=2)   # 16 x 16 windows: 260 tokens
    big.set_hat_operand_dtype("bf16x3")         # r05 refused this geometry ("window has 260 tokens"); r06: fvit_attnlong.hip's two-term instances
    assert big.hat_operand_dtype == "bf16x3" and all(lvl.hat_operand_dtype == "bf16x3" for lvl in big.levels)
    wide = fastervit_amd.create_model("faster_vit_0_224", dim=128, num_heads=[1, 1, 1, 1])   # head_dim 512 / 1024 at the transformer levels
    with pytest.raises(NotImplementedError, match="head_dim"):
        wide.set_hat_operand_dtype("f16x3")
    drop = fastervit_amd.create_model("faster_vit_0_224", attn_drop_rate=0.1)
    with pytest.raises(NotImplementedError, match="attn_drop_rate"):
        drop.set_hat_operand_dtype("f16x3")
