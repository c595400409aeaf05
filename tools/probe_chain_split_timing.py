"""Kernel-level probe (GPU box) of the N > 128 chain route -- per layer avg_heads (HBM stream) + exact-fp32 product (MFMA) -- at the
cfg 1 / 5 / 3 shapes: the product on bmm_f32_tiles.hip vs the general kernel it replaces vs torch.baddbmm (rocBLAS / hipBLASLt),
the head reduction alone, and the whole chain."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer_mm_explainability_amd import ops

FP32_PEAK, HBM_PEAK = 157.3, 8000.0


def us(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


cases = [("square 64 x 1024 (balanced, no padding)", 2, 64, 2, 1024, torch.float32), ("vit-b16 K=32", 12, 32, 12, 197, torch.float32), ("vit-l14@336 B=32 bf16", 24, 32, 16, 577, torch.bfloat16),
         ("detr-enc K=10", 6, 10, 8, 950, torch.float32)]
for name, L, B, H, N, dt in cases:
    attn, grad = [], []
    for _ in range(L):
        a = torch.empty(B * H, N, N, device="cuda", dtype=dt)
        g = torch.empty(B * H, N, N, device="cuda", dtype=dt)
        for i in range(0, B * H, 256):
            a[i:i + 256] = torch.rand(min(256, B * H - i), N, N, device="cuda").softmax(-1).to(dt)
            g[i:i + 256] = (torch.randn(min(256, B * H - i), N, N, device="cuda") * 0.01).to(dt)
        attn.append(a)
        grad.append(g)
    abar = ops.avg_heads(attn[0], grad[0], B)
    R = torch.eye(N, device="cuda").repeat(B, 1, 1) + torch.rand(B, N, N, device="cuda") * 0.01
    flop = 2.0 * B * N ** 3
    slab = 2 * B * H * N * N * attn[0].element_size()
    t_avg = us(lambda: ops.avg_heads(attn[0], grad[0], B))
    print(f"{name}: avg_heads           {t_avg:8.1f} us = {slab / t_avg / 1e3:7.1f} GB/s = {slab / t_avg / 1e3 / HBM_PEAK:.3f} of the HBM peak "
          f"({slab / 1e6:.0f} MB read + {B * N * N * 4 / 1e6:.0f} MB written)")
    for label, opt in (("bmm_f32_tiles (32x32x2)", 1), ("bmm_f32_kernel (general)", 0)):
        ops.set_option("bmm_tiles", opt)
        t = us(lambda: ops.matmul(abar, R, add_to=R))
        print(f"{name}: {label:24s} {t:8.1f} us = {flop / t / 1e6:6.1f} TFLOP/s = {flop / t / 1e6 / FP32_PEAK:.3f} of the fp32 MFMA peak ({flop / 1e9:.2f} GFLOP)")
    ops.set_option("bmm_tiles", 1)
    t = us(lambda: torch.baddbmm(R, abar, R))
    print(f"{name}: torch.baddbmm fp32       {t:8.1f} us = {flop / t / 1e6:6.1f} TFLOP/s = {flop / t / 1e6 / FP32_PEAK:.3f}")
    plan = ops.ChainPlan(attn, grad, B)
    for label in ("chain (avg_heads + product per layer)",):
        t = us(plan.launch, 5)
        print(f"{name}: {label:40s} {t:9.1f} us for {L} layers = {t / L:7.1f} us per layer ({L * (slab + 0.0) / t / 1e3:7.1f} GB/s of slabs, "
              f"{L * flop / t / 1e6:6.1f} TFLOP/s)")
    del attn, grad, plan
    torch.cuda.empty_cache()
