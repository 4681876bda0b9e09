"""Micro-benchmark of the gamma-residual / bias GEMM kernels at the shard-sized shapes of FasterViT-0 (A/B of the tile and ring variants).
usage: python scripts/bench_gemm.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastervit_amd import _lib  # noqa: E402

lib = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
dt, code = torch.float16, 1
SHAPES = [  # name, M, N, K, epilogue (0 bias, 1 gelu, 2 residual)
    ("s3 fc2 shard", 4165, 512, 2048, 2), ("s3 proj shard", 4165, 512, 512, 2), ("s3 fc1 shard", 4165, 2048, 512, 1),
    ("s3 qkv shard", 4165, 1536, 512, 0), ("ct fc2 shard", 1360, 256, 1024, 2), ("ct proj shard", 1360, 256, 256, 2),
    ("ct fc1 shard", 1360, 1024, 256, 1), ("ct qkv shard", 1360, 768, 256, 0),
    ("s3 fc2 full", 12544, 512, 2048, 2), ("s3 fc1 full", 12544, 2048, 512, 1), ("s3 qkv full", 12544, 1536, 512, 0),
    ("s3 proj full", 12544, 512, 512, 2), ("s2 qkv full", 54272, 768, 256, 0), ("s2 fc2 full", 54272, 256, 1024, 2)]
VARIANTS = [("64-row", dict(gemm_bm64_max_grid=400, gemm_stagger=0)),
            ("64-row stagger", dict(gemm_bm64_max_grid=400, gemm_stagger=1)),
            ("128-row", dict(gemm_bm64_max_grid=0, gemm_stagger=0)),
            ("128-row stagger", dict(gemm_bm64_max_grid=0, gemm_stagger=1))]
if len(sys.argv) > 1 and sys.argv[1] == "ring":   # ring depth comes from FVIT_TUNE_gemm_ring in the environment
    VARIANTS = [("64-row", dict(gemm_bm64_max_grid=400, gemm_stagger=0))]
    SHAPES = [x for x in SHAPES if "shard" in x[0]]
if len(sys.argv) > 1 and sys.argv[1] == "old":   # r01 sweep: waves per workgroup x tile height
    VARIANTS = [("8 waves, 64-row", dict(gemm_nw8_max_grid=400, gemm_bm64_max_grid=400)),
                ("8 waves, 128-row", dict(gemm_nw8_max_grid=400, gemm_bm64_max_grid=0)),
                ("4 waves, 64-row", dict(gemm_nw8_max_grid=0, gemm_bm64_max_grid=400)),
                ("4 waves, 128-row", dict(gemm_nw8_max_grid=0, gemm_bm64_max_grid=0))]
if len(sys.argv) > 1 and sys.argv[1] == "fv4":   # r03: the Linear layers of FasterViT-4 (bs 128 / 3 shards and whole batch), 128 x 128 vs 256 x 256 tiles
    VARIANTS = [("128x128", dict(gemm256_min_tiles=0, gemm_pp=0)), ("256x256", dict(gemm256_min_tiles=1, gemm_pp=0)),
                ("256x256 ping-pong", dict(gemm256_min_tiles=1, gemm_pp=1))]
    SHAPES = [("fv4 s2 qkv shard", 9116, 3072, 832, 0), ("fv4 s2 proj shard", 9116, 784, 1024, 2), ("fv4 s2 fc1 shard", 9116, 3136, 832, 1),
              ("fv4 s2 fc2 shard", 9116, 784, 3136, 2), ("fv4 s3 qkv shard", 2107, 6144, 1600, 0), ("fv4 s3 proj shard", 2107, 1568, 2048, 2),
              ("fv4 s3 fc1 shard", 2107, 6272, 1600, 1), ("fv4 s3 fc2 shard", 2107, 1568, 6272, 2),
              ("fv4 s2 qkv full", 27136, 3072, 832, 0), ("fv4 s2 fc1 full", 27136, 3136, 832, 1), ("fv4 s2 fc2 full", 27136, 784, 3136, 2),
              ("fv4 s3 fc1 full", 6272, 6272, 1600, 1), ("fv4 s3 fc2 full", 6272, 1568, 6272, 2),
              ("anyres s2 qkv", 17760, 3072, 832, 0), ("anyres s2 fc2", 17760, 784, 3136, 2), ("8192x8192x1024", 8192, 8192, 1024, 0),
              ("8192x8192x4096", 8192, 8192, 4096, 0), ("4096^3", 4096, 4096, 4096, 0)]
g = torch.Generator(device="cpu").manual_seed(0)
for name, M, N, K, epi in SHAPES:
    Mp = (M + 255) // 256 * 256
    NBUF = 6   # rotate operands so that a launch does not find its own A tile in L2 from the previous launch
    As = [torch.randn(Mp, K, generator=g).to(dt).cuda() for _ in range(NBUF)]
    W = (torch.randn((N + 255) // 256 * 256, K, generator=g) / K ** 0.5).to(dt).cuda()
    bias = torch.zeros(N).cuda()
    gamma = torch.ones(N).cuda()
    X = [torch.zeros(Mp, N).cuda() for _ in range(NBUF)]
    O = [torch.zeros(Mp, N, dtype=dt).cuda() for _ in range(NBUF)]

    def call(i):
        k = i % NBUF
        if epi == 2:
            _lib.check(lib.fvit_gemm_residual(code, As[k].data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), gamma.data_ptr(), X[k].data_ptr(), N,
                                              M, N, K, st), "gemm_residual")
        else:
            _lib.check(lib.fvit_gemm_bias_act(code, As[k].data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), O[k].data_ptr(), N, M, N, K, epi, st),
                       "gemm_bias_act")
    line = f"{name:18s} M={M:5d} N={N:4d} K={K:4d}:"
    for vname, knobs in VARIANTS:
        for k, v in knobs.items():
            _lib.tune(k, v)
        for i in range(3):
            call(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 30
        for i in range(n):
            call(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        line += f"  {vname}: {us:6.1f} us ({2.0 * M * N * K / us / 1e6:6.1f} TF/s)"
    print(line, flush=True)
