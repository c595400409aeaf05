"""Probe (GPU box): exact-fp32 MFMA bmm at the long-sequence chain shapes vs torch.baddbmm (rocBLAS / hipBLASLt fp32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer_mm_explainability_amd import ops

def bench(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

for name, B, N in [("vit-l 32x577", 32, 577), ("detr-enc 10x950", 10, 950), ("vit-b16 32x197", 32, 197), ("4x1024", 4, 1024)]:
    a = torch.rand(B, N, N, device="cuda") / N
    r = torch.rand(B, N, N, device="cuda")
    want = torch.baddbmm(r.double(), a.double(), r.double())
    us = bench(lambda: ops.matmul(a, r, add_to=r))
    err = (ops.matmul(a, r, add_to=r).double() - want).abs().max().item()
    print(f"{name}: mmx bmm: {us:8.1f} us  {2 * B * N ** 3 / us / 1e6:6.1f} TF/s  max err {err:.2e}")
    print(f"{name}: torch.baddbmm fp32: {bench(lambda: torch.baddbmm(r, a, r)):8.1f} us")
