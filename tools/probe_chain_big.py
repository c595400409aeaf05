"""Probe (GPU box): the N > 128 relevancy chain, one-launch persistent team kernel (self_chain_big = 2) vs the per-layer
split path (0), at the cfg 1 / 5 / 3 shapes; prints us per call, algorithmic GB/s (A and G once + R out) and TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer_mm_explainability_amd import ops

def bench(fn, reps=3, iters=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        for _ in range(reps):
            fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (iters * reps) * 1e3

cases = [("vit-b16 K=32", 12, 32, 12, 197, torch.float32), ("vit-l14@336 B=16 bf16", 24, 16, 16, 577, torch.bfloat16),
         ("vit-l14@336 B=32 bf16", 24, 32, 16, 577, torch.bfloat16), ("vit-l14@336 B=64 bf16", 24, 64, 16, 577, torch.bfloat16), ("vit-l14@336 B=128 bf16", 24, 128, 16, 577, torch.bfloat16),
         ("detr-enc K=10", 6, 10, 8, 950, torch.float32),
         ("vit-b16 B=1", 12, 1, 12, 197, torch.float32)]
if len(sys.argv) > 1:
    cases = [c for c in cases if sys.argv[1] in c[0]]
for name, L, B, H, N, dt in cases:
    attn, grad = [], []
    for _ in range(L):       # built per sample block: a [B*H, N, N] fp32 temporary of the largest case would be 27 GB
        a = torch.empty(B * H, N, N, device="cuda", dtype=dt)
        g = torch.empty(B * H, N, N, device="cuda", dtype=dt)
        for i in range(0, B * H, 256):
            a[i:i + 256] = torch.rand(min(256, B * H - i), N, N, device="cuda").softmax(-1).to(dt)
            g[i:i + 256] = (torch.randn(min(256, B * H - i), N, N, device="cuda") * 0.01).to(dt)
        attn.append(a)
        grad.append(g)
    nbytes = 2 * L * B * H * N * N * attn[0].element_size() + B * N * N * 4
    flops = L * B * (2 * H * N * N + 2 * N ** 3)
    outs = {}
    if len(sys.argv) > 2:
        for dbg in (0, 1, 2, 4, 8, 16, 32, 1 | 4, 4 | 8 | 16, 1 | 2 | 4 | 8 | 16 | 32):
            ops.set_option("self_chain_big", 2 | (dbg << 8))
            us = bench(lambda: ops.relevancy_self_chain(attn, grad, B))
            print(f"{name}: team kernel skip={dbg:2d} (1 stream, 2 wait, 4 mfma, 8 staging, 16 Rtile ld/st, 32 slab): {us:9.1f} us", flush=True)
    for mode in (2, 0):
        ops.set_option("self_chain_big", mode)
        us = bench(lambda: ops.relevancy_self_chain(attn, grad, B))
        outs[mode] = ops.relevancy_self_chain(attn, grad, B)
        print(f"{name}: {'team kernel' if mode else 'split path '}: {us:9.1f} us  {nbytes / us / 1e3:7.0f} GB/s  {flops / us / 1e6:6.1f} TF/s", flush=True)
    print(f"{name}: max |team - split| = {(outs[2] - outs[0]).abs().max().item():.3e}")
    del attn, grad
ops.set_option("self_chain_big", 1)
