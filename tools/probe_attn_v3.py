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
modes = list(dict.fromkeys(int(m) for m in sys.argv[2].split(","))) if len(sys.argv) > 2 else [0, 2, 3]
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
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 1
times = {m: [] for m in modes}
stream = torch.cuda.current_stream()
for rnd in range(rounds):                       # interleaved rounds: box / clock drift hits every mode alike
    for mode in modes:
        ops.set_option("attn_bf16_v3", mode)
        fn = lambda: ops.attn_capture_bwd(q, k, v, probs, d_o, None, D ** -0.5, batch=B, o=o,  # noqa: E731
                                          out=(out[:, :, 0], out[:, :, 1], out[:, :, 2]), mma_bf16=True, rel_row=rel)
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 8
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        times[mode].append(e0.elapsed_time(e1) / reps * 1e3)
for mode in modes:
    ts = sorted(times[mode])
    us = ts[len(ts) // 2]
    print("attn_bf16_v3 = %2d : median %.1f us (min %.1f, max %.1f, %d rounds) per layer at B = %d = %.1f TFLOP/s algorithmic (%.1f %% of 2.5 PFLOP/s); executed %.1f TFLOP/s (%.1f %%)"
          % (mode, us, ts[0], ts[-1], len(ts), B, flop / us / 1e6, flop / us / 1e6 / 2500 * 100, flop_exec / us / 1e6, flop_exec / us / 1e6 / 2500 * 100), flush=True)
