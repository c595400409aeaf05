"""Kernel-only probe (GPU box): the fused chain kernel at the cfg-2 shapes over ROTATING slab sets (> 600 MB per tower, so
neither tower is served from the 256 MiB Infinity Cache), default cache policy vs nt loads (option "self_chain_nt")."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer_mm_explainability_amd import ops


def timed(fns, iters=10, warm=2):
    for _ in range(warm):
        for f in fns:
            f()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for f in fns:
            f()
    graph.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        graph.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (iters * len(fns)) * 1e3


def main():
    B = 64
    for (L, H, N, name, sets) in [(12, 8, 77, "txt", 3), (12, 12, 50, "img", 4)]:
        slabs = []
        for _ in range(sets):
            attn = [torch.rand(B * H, N, N, device="cuda").softmax(-1) for _ in range(L)]
            grad = [torch.randn(B * H, N, N, device="cuda") * 0.01 for _ in range(L)]
            slabs.append((attn, grad))
        nbytes = 2 * L * B * H * N * N * 4
        outs = []
        for nt in (0, 1, 0, 1):
            ops.set_option("self_chain_nt", nt)
            rot = timed([(lambda a=a, g=g: ops.relevancy_self_chain(a, g, B)) for a, g in slabs])
            same = timed([lambda: ops.relevancy_self_chain(slabs[0][0], slabs[0][1], B)] * sets)
            outs.append(ops.relevancy_self_chain(slabs[0][0], slabs[0][1], B).clone())
            print(f"{name}: nt={nt} rotating {sets} x {nbytes/1e6:.0f} MB: {rot:.1f} us = {nbytes/rot/1e6:.3f} TB/s | "
                  f"same buffers: {same:.1f} us = {nbytes/same/1e6:.3f} TB/s")
        print(f"{name}: nt vs default bit-identical: {bool((outs[0] == outs[1]).all())}")
    ops.set_option("self_chain_nt", 0)


if __name__ == "__main__":
    main()
