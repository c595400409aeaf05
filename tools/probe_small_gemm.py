"""Probe (GPU box): the library's fp32 GEMM vs our exact-fp32 MFMA bmm kernel at the small / skinny shapes of the LXMERT (B = 32:
448 text rows, 1152 region rows) and DETR (100 queries) bodies."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformer_mm_explainability_amd import ops  # noqa: E402


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for M, K, N in [(448, 768, 768), (448, 768, 2304), (448, 768, 3072), (448, 3072, 768), (1152, 768, 768), (1152, 768, 2304),
                (1152, 768, 3072), (1152, 3072, 768), (100, 256, 256), (100, 256, 2048), (100, 2048, 256), (950, 2048, 256),
                (950, 256, 2048), (1000, 256, 256)]:
    a, b = torch.randn(M, K, device="cuda"), torch.randn(K, N, device="cuda")
    t_lib = timed(lambda: torch.mm(a, b))
    t_own = timed(lambda: ops.matmul(a, b))
    err = float((ops.matmul(a, b) - torch.mm(a, b)).abs().max())
    print("M=%5d K=%5d N=%5d  library %6.1f us (%5.1f TF/s) | bmm_f32_kernel %6.1f us (%5.1f TF/s)  max|diff| %.1e"
          % (M, K, N, t_lib, 2 * M * K * N / t_lib / 1e6, t_own, 2 * M * K * N / t_own / 1e6, err), flush=True)
