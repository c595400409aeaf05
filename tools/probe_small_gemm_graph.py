"""In-graph (hipGraph replay: no host launch cost) time of the one-sample forward GEMMs of the DETR encoder (M = 950 rows) and
decoder (M = 100): library F.linear against our exact-fp32 MFMA bmm kernel on the cached transposed weight (+ a bias add)."""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from transformer_mm_explainability_amd import ops  # noqa: E402


def graph_us(fn, n=40):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * n) * 1e3


print("%-20s %12s %14s %16s" % ("M, K, N", "F.linear us", "bmm_f32 us", "bmm_f32 + bias us"))
for M, K, N in [(950, 256, 512), (950, 256, 256), (950, 256, 2048), (950, 2048, 256), (950, 256, 1536), (100, 256, 512),
                (100, 256, 256), (100, 256, 2048), (100, 2048, 256), (448, 768, 768), (448, 768, 2304), (448, 768, 3072),
                (448, 3072, 768), (1152, 768, 768), (1152, 768, 2304), (1152, 768, 3072), (1152, 3072, 768)]:
    x = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") * 0.05
    b = torch.randn(N, device="cuda")
    Wt = W.t().contiguous()
    t_lib = graph_us(lambda: F.linear(x, W, b))
    t_own = graph_us(lambda: ops.matmul(x, Wt))
    t_own_b = graph_us(lambda: ops.matmul(x, Wt).add_(b))
    err = float((ops.matmul(x, Wt) + b - F.linear(x, W, b)).abs().max())
    print("%-20s %12.1f %14.1f %16.1f   max|diff| %.1e" % ((M, K, N), t_lib, t_own, t_own_b, err))
