"""Quick kernel-only probe (GPU box): relevancy chain at CLIP ViT-B/32 shapes, HIP-event timed."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer_mm_explainability_amd import ops

def bench(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us

B = int(os.environ.get("B", 64))
for (L, H, N, name) in [(12, 12, 50, "img"), (12, 8, 77, "txt"), (12, 12, 197, "vitb16")]:
    attn = [torch.rand(B * H, N, N, device="cuda").softmax(-1) for _ in range(L)]
    grad = [torch.randn(B * H, N, N, device="cuda") * 0.01 for _ in range(L)]
    nbytes = 2 * L * B * H * N * N * 4
    for algo in (1, 2):
        ops.set_option("self_chain_algo", algo)
        us = bench(lambda: ops.relevancy_self_chain(attn, grad, B))
        print(f"{name}: chain algo={algo} B={B} L={L} H={H} N={N}: {us:.1f} us  {nbytes/us/1e6:.3f} TB/s ({nbytes/1e6:.1f} MB)")
    ops.set_option("self_chain_algo", 2)
    for dbg in (1, 2):
        ops.set_option("debug_flags", dbg)
        us = bench(lambda: ops.relevancy_self_chain(attn, grad, B))
        print(f"{name}: chain algo=2 debug={dbg} ({'stream only' if dbg == 1 else 'ticket+chain only'}): {us:.1f} us")
    ops.set_option("debug_flags", 0)
    ops.set_option("self_chain_algo", 0)
    us2 = bench(lambda: [ops.avg_heads(a, g, B) for a, g in zip(attn, grad)])
    print(f"{name}: avg_heads x{L}: {us2:.1f} us  {nbytes/us2/1e6:.3f} TB/s")
