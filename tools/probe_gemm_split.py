"""Probe (GPU box): would a split-bf16 GEMM (fp32 operands as 3 bf16 pieces each, products concatenated along K, fp32 accumulate /
output: ONE library bf16 GEMM with K' = 3K or 6K) beat the library's exact-fp32 GEMM at the cfg-2 body shapes?  """
import time

import torch

shapes = [(4928, 512, 1536), (4928, 512, 512), (4928, 512, 2048), (4928, 2048, 512),
          (3200, 768, 2304), (3200, 768, 768), (3200, 768, 3072), (3200, 3072, 768)]


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


tot = {"fp32": 0.0, "x3": 0.0, "x6": 0.0}
for M, K, N in shapes:
    a, b = torch.randn(M, K, device="cuda"), torch.randn(K, N, device="cuda")
    t32 = timed(lambda: torch.mm(a, b))
    row = "M=%5d K=%5d N=%5d  fp32 %7.1f us (%5.1f TF/s)" % (M, K, N, t32, 2 * M * K * N / t32 / 1e6)
    tot["fp32"] += t32
    for mult in (3, 6):
        a6 = torch.randn(M, mult * K, device="cuda").to(torch.bfloat16)
        b6 = torch.randn(mult * K, N, device="cuda").to(torch.bfloat16)
        t = timed(lambda: torch.mm(a6, b6, out_dtype=torch.float32))
        tot["x%d" % mult] += t
        row += " | bf16 K'=%dK %7.1f us (%6.1f TF/s bf16)" % (mult, t, 2 * M * mult * K * N / t / 1e6)
    print(row, flush=True)
print("sum: fp32 %.1f us, bf16x3 %.1f us, bf16x6 %.1f us (GEMM only; the split pass over the activations comes on top)"
      % (tot["fp32"], tot["x3"], tot["x6"]))
