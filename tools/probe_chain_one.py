"""Run the text-tower chain launch (B=64, L=12, H=8, N=77, auto groups) a few times -- target for rocprofv3 --pmc."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer_mm_explainability_amd import ops
B, L, H, N = 64, 12, 8, 77
attn = [torch.rand(B * H, N, N, device="cuda").softmax(-1) for _ in range(L)]
grad = [torch.randn(B * H, N, N, device="cuda") * 0.01 for _ in range(L)]
plan = ops.ChainPlan(attn, grad, B)
for _ in range(6):
    plan.launch()
torch.cuda.synchronize()
