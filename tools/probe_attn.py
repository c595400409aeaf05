"""Probe (GPU box): attention-capture forward/backward at CLIP shapes, hipGraph-replay timed, with phase-skip flags."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer_mm_explainability_amd import ops
from tools.probe_chain import bench  # noqa

for (B, H, N, D, name) in [(64, 12, 50, 64, "img"), (64, 8, 77, 64, "txt")]:
    qkv = torch.randn(B, N, 3, H, D, device="cuda")
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    probs = torch.empty(B, H, N, N, device="cuda"); dprobs = torch.empty_like(probs)
    d_o = torch.randn(B, N, H, D, device="cuda")
    mask = torch.full((N, N), float("-inf"), device="cuda").triu_(1) if name == "txt" else None
    for flags in (0, 1, 2, 4, 8, 15):
        ops.set_option("attn_small", 1 | (flags << 8))
        us = bench(lambda: ops.attn_capture_fwd(q, k, v, probs, D ** -0.5, 0, mask))
        print(f"{name}: fwd small skip={flags:2d} (1=S 2=softmax 4=PV 8=stage): {us:.1f} us")
    ops.set_option("attn_small", 0)
    us = bench(lambda: ops.attn_capture_fwd(q, k, v, probs, D ** -0.5, 0, mask))
    print(f"{name}: fwd tiled: {us:.1f} us")
    for small in (1, 0):
        ops.set_option("attn_small", small)
        ops.attn_capture_fwd(q, k, v, probs, D ** -0.5, 0, mask)
        us = bench(lambda: ops.attn_capture_bwd(q, k, v, probs, d_o, dprobs, D ** -0.5, 0))
        print(f"{name}: bwd small={small}: {us:.1f} us")
ops.set_option("attn_small", 1)

# access-pattern A/B: same kernel on contiguous [B,H,N,D] operands
for (B, H, N, D, name) in [(64, 12, 50, 64, "img")]:
    q = torch.randn(B, H, N, D, device="cuda"); k = torch.randn_like(q); v = torch.randn_like(q)
    probs = torch.empty(B, H, N, N, device="cuda")
    for flags in (0, 8):
        ops.set_option("attn_small", 1 | (flags << 8))
        us = bench(lambda: ops.attn_capture_fwd(q, k, v, probs, D ** -0.5, 0, None, layout="bhnd"))
        print(f"{name}: fwd small CONTIGUOUS bhnd skip={flags}: {us:.1f} us")
ops.set_option("attn_small", 1)
