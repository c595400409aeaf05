"""Probe (GPU box): LXMERT rule schedule, fused single launch vs per-rule kernels, wall time per sample."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_generators import lxmert_usage
from transformer_mm_explainability_amd import lxmert_explainability as le
g = dict(np.load("tests/golden/lxmert_chain_full.npz"))
usage = lxmert_usage(g)
for fused in (True, False):
    gen = le.GeneratorOurs(usage); gen.fused = fused
    for _ in range(3): gen.generate_ours(None, use_lrp=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): gen.generate_ours(None, use_lrp=False)
    torch.cuda.synchronize()
    print(f"lxmert generate_ours (T=14, I=36, 9+5+5 layers) fused={fused}: {(time.perf_counter()-t0)/20*1e6:.0f} us per sample (incl. host)")
