"""Streaming attention backward at the ViT-L/14@336 shape (577 tokens, 16 heads x 64, shared forward) in its modes:
exact fp32 MFMA / bf16 MFMA, fp32 or bf16 gradient I/O, with and without the row-relevancy reduction (no dP slab)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from transformer_mm_explainability_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N, H, D = 577, 16, 64
dev = "cuda"
g = torch.Generator().manual_seed(0)
qkv = torch.randn(1, N, 3, H, D, generator=g).to(dev)
q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
d_o32 = (torch.randn(B, N, H, D, generator=g) * 1e-2).to(dev)
row = torch.zeros(B, N, device=dev)
row[:, 0] = 1


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for slab in (torch.float32, torch.bfloat16):
    probs = torch.empty(1, H, N, N, device=dev, dtype=slab)
    o = ops.attn_capture_fwd(q, k, v, probs, D ** -0.5, mma_bf16=slab == torch.bfloat16)
    dprobs = torch.empty(B, H, N, N, device=dev, dtype=slab)
    for mma, io16, rel in ((False, False, False), (True, False, False), (True, True, False), (True, False, True), (True, True, True)):
        d_o = d_o32.to(torch.bfloat16) if io16 else d_o32
        out = torch.empty(B, N, 3, H, D, device=dev, dtype=d_o.dtype)
        outs = (out[:, :, 0], out[:, :, 1], out[:, :, 2])
        ms = timed(lambda: ops.attn_capture_bwd(q, k, v, probs, d_o, None if rel else dprobs, D ** -0.5, batch=B, o=o, out=outs,
                                                mma_bf16=mma, rel_row=row if rel else None))
        print("slab %-8s  %s MFMA  grad i/o %s  %s: %7.3f ms per layer at B = %d"
              % (str(slab).replace("torch.", ""), "bf16" if mma else "fp32", "bf16" if io16 else "fp32",
                 "row relevancy (no dP slab)" if rel else "dP slab written        ", ms, B))
