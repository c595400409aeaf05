"""rocprofv3 target: the DEFAULT N > 128 chain (split path: avg_heads + exact-fp32 bmm per layer) at the cfg 1 / 5 / 3 shapes,
5 calls each (for --kernel-trace --stats and the separate --pmc FETCH_SIZE / WRITE_SIZE passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer_mm_explainability_amd import ops

cases = [("vit-b16 K=32", 12, 32, 12, 197, torch.float32), ("vit-l14@336 B=32 bf16", 24, 32, 16, 577, torch.bfloat16),
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
    for _ in range(5):
        ops.relevancy_self_chain(attn, grad, B)
    torch.cuda.synchronize()
    slab = 2 * B * H * N * N * attn[0].element_size()
    print(f"{name}: per layer  avg_heads reads {slab / 1e6:.1f} MB + writes {B * N * N * 4 / 1e6:.1f} MB;  bmm reads 2 x {B * N * N * 4 / 1e6:.1f} MB "
          f"+ writes {B * N * N * 4 / 1e6:.1f} MB, {2 * B * N ** 3 / 1e9:.2f} GFLOP")
    del attn, grad
