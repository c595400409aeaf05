"""GPU box: a few launches of the one-launch long-sequence layer kernel (cfg 5's variant shape) for rocprofv3 counter passes.
argv[1] = debug_flags (0 full, 32 product only, 64 head reduction only)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformer_mm_explainability_amd import ops
dbg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B, H, N, L = 128, 16, 577, 3
g = torch.Generator(device="cuda").manual_seed(0)
attn = [torch.rand(H, N, N, device="cuda", generator=g).softmax(-1).to(torch.bfloat16)] * L
grad = [(torch.randn(B * H, N, N, device="cuda", generator=g) * 0.05).to(torch.bfloat16)] * L
ops.set_option("self_chain_rows", 1)
ops.set_option("debug_flags", dbg)
for _ in range(2):
    ops.relevancy_self_chain(attn, grad, B, shared_attn=True)
torch.cuda.synchronize()
