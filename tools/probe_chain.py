"""Kernel-only probe (GPU box): relevancy chain at CLIP ViT-B/32 shapes, HIP-event timed.
Prints per-tower timings for every algorithm / layer-group setting and the two towers launched concurrently."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer_mm_explainability_amd import ops


def bench(fn, iters=20, warm=3):
    """Time `fn` replayed from a captured hipGraph (no Python / ctypes launch overhead in the timed region)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    try:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            fn()
        fn = graph.replay
        fn()
    except Exception as exc:  # noqa: BLE001
        print("graph capture failed, timing eager launches:", exc)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def main():
    B = int(os.environ.get("B", 64))
    towers = {}
    for (L, H, N, name) in [(12, 12, 50, "img"), (12, 8, 77, "txt")]:
        attn = [torch.rand(B * H, N, N, device="cuda").softmax(-1) for _ in range(L)]
        grad = [torch.randn(B * H, N, N, device="cuda") * 0.01 for _ in range(L)]
        towers[name] = (attn, grad, 2 * L * B * H * N * N * 4)
        nbytes = towers[name][2]
        for algo, groups in ((1, 1), (1, 2), (1, 3), (1, 4), (1, 6), (2, 1)):
            ops.set_option("self_chain_algo", algo)
            ops.set_option("self_chain_groups", groups)
            us = bench(lambda: ops.relevancy_self_chain(attn, grad, B))
            print(f"{name}: chain algo={algo} groups={groups} B={B} L={L} H={H} N={N}: {us:.1f} us  {nbytes/us/1e6:.3f} TB/s ({nbytes/1e6:.1f} MB)")
    ops.set_option("self_chain_algo", 1)
    for name in ("img", "txt"):
        attn, grad, nbytes = towers[name]
        for groups in (1, 4):
            for dbg in (1, 4, 5):
                ops.set_option("self_chain_groups", groups)
                ops.set_option("debug_flags", dbg)
                us = bench(lambda: ops.relevancy_self_chain(attn, grad, B))
                print(f"{name}: algo=1 groups={groups} debug={dbg} (1=no combine, 4=no mfma): {us:.1f} us  {nbytes/us/1e6:.3f} TB/s")
    ops.set_option("debug_flags", 0)
    side = torch.cuda.Stream()
    total = towers["img"][2] + towers["txt"][2]
    for gi, gt in ((1, 1), (1, 2), (1, 3), (2, 2), (2, 3), (2, 4), (3, 3), (4, 4)):
        def both():
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                ops.set_option("self_chain_groups", gt)
                ops.relevancy_self_chain(towers["txt"][0], towers["txt"][1], B)
            ops.set_option("self_chain_groups", gi)
            ops.relevancy_self_chain(towers["img"][0], towers["img"][1], B)
            cur.wait_stream(side)
        us = bench(both)
        print(f"both towers concurrently, groups img={gi} txt={gt}: {us:.1f} us  {total/us/1e6:.3f} TB/s ({total/1e6:.1f} MB)")
    ops.set_option("self_chain_groups", 0)
    ops.set_option("self_chain_algo", 0)


if __name__ == "__main__":
    main()
