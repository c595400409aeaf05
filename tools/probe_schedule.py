"""Kernel-only probe (GPU box): the bi-modal rule schedule at the cfg-4 shapes (LXMERT-base: 9 / 5 / 5 layers, 12 heads, T = 14,
I = 36) -- one workgroup per sample (rounds 1-3, algo 1) vs the two-phase kernel (round 4, algo 2), HIP-event timed from a
replayed hipGraph, over rotating slab sets."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer_mm_explainability_amd import ops


def timed(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        fn()
    graph.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        graph.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    H, T, I = 12, 14, 36
    for B in (1, 32, 128):
        sm = lambda *s: torch.softmax(torch.randn(*s, device="cuda"), -1)
        gr = lambda *s: torch.randn(*s, device="cuda") * 1e-2
        pair = lambda nq, nk: (sm(B, H, nq, nk), gr(B, H, nq, nk))
        groups = ([pair(T, T) for _ in range(9)], [pair(I, I) for _ in range(5)], [pair(T, I) for _ in range(5)],
                  [pair(I, T) for _ in range(5)], [pair(T, T) for _ in range(5)], [pair(I, I) for _ in range(5)])
        nbytes = sum(a.numel() * 8 for grp in groups for a, _ in grp) + B * (T * T + T * I + I * I + I * T) * 4
        us = timed(lambda: ops.lxmert_schedule(*groups, check_diag="defer"))
        print(f"B={B:4d}: {us:8.1f} us  {nbytes/us/1e3:8.1f} GB/s ({nbytes/1e6:.1f} MB)  {us/B:6.2f} us/sample")
        for dbg, what in ((1, "phase 1 only"), (2, "no MFMA tiles"), (62, "skeleton only")):
            os.environ["MMX_BM_DEBUG"] = str(dbg)
            us = timed(lambda: ops.lxmert_schedule(*groups, check_diag="defer"))
            print(f"B={B:4d} debug={dbg} ({what}): {us:8.1f} us")
        os.environ.pop("MMX_BM_DEBUG")
        os.environ["MMX_BM_SPLIT"] = "1"
        us = timed(lambda: ops.lxmert_schedule(*groups, check_diag="defer"))
        print(f"B={B:4d} two launches (phase 1: one workgroup per block; phase 2: one per sample): {us:8.1f} us")
        os.environ["MMX_BM_SPLIT"] = "0"


if __name__ == "__main__":
    main()
