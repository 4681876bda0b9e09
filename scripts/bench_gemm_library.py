"""What does the library GEMM (hipBLASLt through torch.matmul, fp16, fp32 accumulate) reach on the FasterViT-4 Linear shapes?  A ceiling check for
gemm_kernel (scripts/bench_gemm.py fv4 times ours on the same shapes); not used by the product."""
import torch

shapes = [("fv4 s2 qkv shard", 9116, 3072, 832), ("fv4 s2 proj shard", 9116, 784, 1024), ("fv4 s2 fc1 shard", 9116, 3136, 832),
          ("fv4 s2 fc2 shard", 9116, 784, 3136), ("fv4 s3 qkv shard", 2107, 6144, 1600), ("fv4 s3 proj shard", 2107, 1568, 2048),
          ("fv4 s3 fc1 shard", 2107, 6272, 1600), ("fv4 s3 fc2 shard", 2107, 1568, 6272),
          ("fv4 s2 fc1 batch", 27136, 3136, 832), ("fv4 s2 fc2 batch", 27136, 784, 3136), ("fv4 s3 fc1 batch", 6272, 6272, 1600),
          ("fv4 s3 fc2 batch", 6272, 1568, 6272), ("square 8192", 8192, 8192, 8192)]
g = torch.Generator(device="cpu").manual_seed(0)
for name, M, N, K in shapes:
    xs = [torch.randn(M, K, generator=g).half().cuda() for _ in range(4)]
    ws = [torch.randn(N, K, generator=g).half().cuda() for _ in range(4)]
    out = torch.empty(M, N, dtype=torch.float16, device="cuda")
    for i in range(3):
        torch.matmul(xs[i % 4], ws[i % 4].t(), out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(20):
        torch.matmul(xs[i % 4], ws[i % 4].t(), out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"{name:20s} M={M:6d} N={N:5d} K={K:5d}: library {us:8.1f} us ({2.0 * M * N * K / us / 1e6:7.1f} TF/s)", flush=True)
