"""Probe (GPU box): phase-skip timing of attn_fwd_head_kernel (profiling flags in the attn_head option: bits 8..15)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer_mm_explainability_amd import ops
from tools.probe_attn_head import bench  # noqa

for (B, H, N, D, name) in [(64, 8, 77, 64, "txt")]:
    for layout in ("bnhd", "bhnd"):
        if layout == "bnhd":
            qkv = torch.randn(B, N, 3, H, D, device="cuda")
            q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
        else:
            q, k, v = (torch.randn(B, H, N, D, device="cuda") for _ in range(3))
        probs = torch.empty(B, H, N, N, device="cuda")
        for skip in (0, 1, 2, 4, 8, 16, 4 | 16, 2 | 8, 1 | 4 | 16, 1 | 2 | 8, 31):
            ops.set_option("attn_head", 1 | (skip << 8))
            us = bench(lambda: ops.attn_capture_fwd(q, k, v, probs, D ** -0.5, 0, None, layout=layout))
            print(f"{name} {layout}: fwd skip={skip:2d} (1 loads, 2 S, 4 Pstore, 8 PV, 16 Ostore): {us:6.1f} us")
ops.set_option("attn_head", 1)
