"""Small-M fp32 GEMMs of the one-image / small-batch bodies: library (torch.mm / addmm) vs the option "linear_stream" kernel
(csrc/linear_stream.hip), both replayed from a hipGraph of launches that rotate over 16 distinct weight matrices (eager timing
below ~19 us measures the host's launch rate)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from transformer_mm_explainability_amd import ops  # noqa: E402

SETS = 16


def graph_us(fn, n_launches, reps=10):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with ops.graph_capture(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps / n_launches * 1e3


shapes = [("vit-b16 qkv", 197, 768, 2304), ("vit-b16 proj", 197, 768, 768), ("vit-b16 fc1", 197, 768, 3072), ("vit-b16 fc2", 197, 3072, 768),
          ("lxmert lang 768", 448, 768, 768), ("lxmert lang fc1", 448, 768, 3072), ("lxmert lang fc2", 448, 3072, 768),
          ("lxmert visn 768", 1152, 768, 768), ("lxmert visn qkv", 1152, 768, 2304), ("lxmert visn fc1", 1152, 768, 3072),
          ("lxmert visn fc2", 1152, 3072, 768), ("detr dec 256", 100, 256, 256), ("detr dec ffn1", 100, 256, 2048),
          ("detr dec ffn2", 100, 2048, 256), ("detr enc 256", 950, 256, 256), ("detr enc ffn1", 950, 256, 2048),
          ("detr enc ffn2", 950, 2048, 256)]
print("%-18s %5s %5s %5s | %9s %9s | %9s %7s | max|diff|" % ("shape", "M", "K", "N", "library", "TF/s", "stream", "TF/s"))
for name, M, K, N in shapes:
    x = torch.randn(M, K, device="cuda") / K ** 0.5
    ws = [torch.randn(K, N, device="cuda") for _ in range(SETS)]
    outs = [torch.empty(M, N, device="cuda") for _ in range(SETS)]
    t_lib = graph_us(lambda: [torch.mm(x, w, out=o) for w, o in zip(ws, outs)], SETS)
    ref = outs[0].clone()
    ops.set_option("linear_stream", 1)
    try:
        t_new = graph_us(lambda: [ops.matmul(x, w) for w in ws], SETS)
        got = ops.matmul(x, ws[0])
    finally:
        ops.set_option("linear_stream", 0)
    fl = 2.0 * M * K * N
    print("%-18s %5d %5d %5d | %7.1f us %7.1f   | %7.1f us %7.1f | %.1e" % (name, M, K, N, t_lib, fl / t_lib / 1e6, t_new, fl / t_new / 1e6,
                                                                               float((got - ref).abs().max())))
