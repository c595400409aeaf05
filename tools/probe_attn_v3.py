"""Probe (GPU box): the cfg-5 attention backward pair of ONE image-tower layer (16 heads x 577 tokens x 64, shared forward, bf16
gradient stream, row-relevancy mode) -- second generation (attn_bf16_v3 = 0) vs third generation (= 2, the default).  FLOPs: ALGORITHMIC (4 products: dP, dV, dQ, dK) and executed
(5: both kernels of the pair compute dP)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformer_mm_explainability_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
modes = [int(m) for m in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 2]
H, N, D = 16, 577, 64
qkv = torch.randn(1, N, 3, H, D, device="cuda")
q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
d_o = (torch.randn(B, N, H, D, device="cuda") * 1e-2).to(torch.bfloat16)
probs = torch.empty(1, H, N, N, device="cuda", dtype=torch.bfloat16)
o = ops.attn_capture_fwd(q, k, v, probs, D ** -0.5, mma_bf16=True)
out = torch.empty(B, N, 3, H, D, device="cuda", dtype=torch.bfloat16)
rel = torch.zeros(B, N, device="cuda")
rel[:, 0] = 1
flop, flop_exec = 4 * 2 * B * H * N * N * D, 5 * 2 * B * H * N * N * D
for mode in modes:
    ops.set_option("attn_bf16_v3", mode)
    fn = lambda: ops.attn_capture_bwd(q, k, v, probs, d_o, None, D ** -0.5, batch=B, o=o,  # noqa: E731
                                      out=(out[:, :, 0], out[:, :, 1], out[:, :, 2]), mma_bf16=True, rel_row=rel)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / reps * 1e6
    print("attn_bf16_v3 = %d : %.1f us per layer at B = %d = %.1f TFLOP/s algorithmic (%.1f %% of 2.5 PFLOP/s); executed %.1f TFLOP/s (%.1f %%)"
          % (mode, us, B, flop / us / 1e6, flop / us / 1e6 / 2500 * 100, flop_exec / us / 1e6, flop_exec / us / 1e6 / 2500 * 100), flush=True)
