"""Probe (GPU box): how fast can the captured-slab layout be streamed at all?  One avg_heads launch over all layers
(pure head reduction, no chain) and a plain torch copy / sum of the same bytes, hipGraph-replay timed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer_mm_explainability_amd import ops
from tools.probe_chain import bench  # noqa: E402

for (L, B, H, N, name) in [(12, 64, 8, 77, "txt"), (12, 64, 12, 50, "img"), (12, 64, 8, 80, "txt-aligned N=80")]:
    A = torch.rand(L * B * H, N, N, device="cuda"); G = torch.randn(L * B * H, N, N, device="cuda")
    nbytes = 2 * A.numel() * 4
    us = bench(lambda: ops.avg_heads(A, G, batch_size=L * B))
    print(f"{name}: avg_heads one launch over {nbytes/1e6:.0f} MB: {us:.1f} us  {nbytes/us/1e6:.2f} TB/s")
    out = torch.empty_like(A)
    us = bench(lambda: torch.add(A, G, out=out))
    print(f"{name}: torch.add(A, G) (reads 2x, writes 1x = {1.5*nbytes/1e6:.0f} MB): {us:.1f} us  {1.5*nbytes/us/1e6:.2f} TB/s")
