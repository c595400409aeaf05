import os, sys
sys.path.insert(0, os.getcwd())
import torch
from transformer_mm_explainability_amd import ops
def timed(f, iters=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (BL, H, N, name) in [(768, 8, 77, "txt all layers as one batch"), (768, 12, 50, "img all layers as one batch"), (64, 8, 77, "txt one layer"), (2048, 8, 77, "txt x2.7")]:
    sets = []
    for _ in range(3 if BL <= 768 else 1):
        sets.append((torch.rand(BL * H, N, N, device="cuda"), torch.randn(BL * H, N, N, device="cuda")))
    nbytes = 2 * BL * H * N * N * 4
    k = [0]
    def f():
        a, g = sets[k[0] % len(sets)]; k[0] += 1
        ops.avg_heads(a, g, BL)
    us = timed(f)
    print(f"avg_heads {name}: B={BL} H={H} N={N}: {us:.1f} us = {nbytes/us/1e6:.3f} TB/s read ({nbytes/1e6:.0f} MB) + {BL*N*N*4/1e6:.0f} MB written")
