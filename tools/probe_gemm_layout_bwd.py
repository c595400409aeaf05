"""Input-gradient GEMMs of the headline step (CLIP ViT-B/32, batch 64, fp32): dX = dY . W in the library's NN layout (torch.matmul(dY, W),
what ops.backward_gemm runs) vs the same product presented as F.linear(dY, W^T contiguous) -- the layout of the FORWARD GEMMs.
    python tools/probe_gemm_layout_bwd.py [tuned]      ("tuned": with the shipped TunableOp selection, as bench.py runs)"""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if len(sys.argv) > 1 and sys.argv[1] == "tuned":
    from transformer_mm_explainability_amd import tuned_gemms
    print("tuned selection loaded:", tuned_gemms.enable("clip_vitb32_b64"))

SHAPES = [("text out_proj", 4928, 512, 512), ("text in_proj", 4928, 1536, 512), ("text c_proj", 4928, 512, 2048), ("text c_fc", 4928, 2048, 512),
          ("image out_proj", 3200, 768, 768), ("image in_proj", 3200, 2304, 768), ("image c_proj", 3200, 768, 3072), ("image c_fc", 3200, 3072, 768)]


def us(fn, reps=60):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            fn()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps // 10):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps // 10 * 10) * 1e6


tot = [0.0, 0.0]
for name, rows, n_out, n_in in SHAPES:               # W [out, in]; dY [rows, out] -> dX [rows, in]
    W = torch.randn(n_out, n_in, device="cuda")
    Wt = W.t().contiguous()
    dY = torch.randn(rows, n_out, device="cuda")
    a = us(lambda: torch.matmul(dY, W))
    b = us(lambda: F.linear(dY, Wt))
    err = float((torch.matmul(dY, W) - F.linear(dY, Wt)).abs().max())
    flop = 2.0 * rows * n_out * n_in
    tot[0] += a
    tot[1] += b
    print("%-16s rows %5d  out %5d -> in %5d | NN matmul %7.1f us (%5.1f TF/s) | F.linear(dY, W^T) %7.1f us (%5.1f TF/s) | max diff %.1e"
          % (name, rows, n_out, n_in, a, flop / a / 1e6, b, flop / b / 1e6, err), flush=True)
print("sum per layer pair: NN %.1f us, NT %.1f us" % tuple(tot))
