"""rocprofv3 target: only the large exact-fp32 products (bmm_f32_tiles.hip) at the chain shapes, 5 launches each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer_mm_explainability_amd import ops
for B, N in ((32, 577), (10, 950), (32, 197), (64, 1024)):
    a = torch.rand(B, N, N, device="cuda") / N
    r = torch.rand(B, N, N, device="cuda")
    for _ in range(5):
        ops.matmul(a, r, add_to=r)
    torch.cuda.synchronize()
