"""GPU box: the long-sequence chain (N > 128) as ONE launch per layer (relevancy_chain_rows.hip, option self_chain_rows = 1) against
the two-launch form (avg_heads_kernel + tiled product, = 0): time per chain, interleaved rounds, and the largest difference of the
results.  Shapes: cfg 5's variant (577 tokens, 16 heads, bf16 slabs, shared probabilities, B = 128), the same in fp32 at B = 32,
DETR's encoder (950 tokens, 8 heads, 6 layers, K = 10), ViT-B/16 (197 tokens, 12 heads, 12 layers, B = 8)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformer_mm_explainability_amd import ops

CASES = [("cfg5 variant: 577 tok, H16, bf16, shared A, B128, 24 layers", 24, 128, 16, 577, torch.bfloat16, True),
         ("577 tok, H16, fp32, per-sample A, B32, 4 layers", 4, 32, 16, 577, torch.float32, False),
         ("DETR encoder: 950 tok, H8, fp32, B10, 6 layers", 6, 10, 8, 950, torch.float32, False),
         ("ViT-B/16: 197 tok, H12, fp32, B8, 12 layers", 12, 8, 12, 197, torch.float32, False)]
only = int(sys.argv[1]) if len(sys.argv) > 1 else None
for ci, (name, L, B, H, N, dt, shared) in enumerate(CASES):
    if only is not None and ci != only:
        continue
    g = torch.Generator(device="cuda").manual_seed(ci)
    # the layers share ONE pair of slabs per case (timing: the kernels read the same number of bytes; rotating sets for > L3 sizes)
    sets = 2 if B * H * N * N * dt.itemsize > 200e6 else 1
    attn = [torch.rand((1 if shared else B) * H, N, N, device="cuda", generator=g).softmax(-1).to(dt) for _ in range(sets)]
    grad = [(torch.randn(B * H, N, N, device="cuda", generator=g) * 0.05).to(dt) for _ in range(sets)]
    al, gl = [attn[l % sets] for l in range(L)], [grad[l % sets] for l in range(L)]
    res, times = {}, {0: [], 1: []}
    for rnd in range(3):
        for mode in (0, 1):
            ops.set_option("self_chain_rows", mode)
            fn = lambda: ops.relevancy_self_chain(al, gl, B, shared_attn=shared)
            out = fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                fn()
            e1.record()
            torch.cuda.synchronize()
            times[mode].append(e0.elapsed_time(e1) / 3)
            res[mode] = out
    ops.set_option("self_chain_rows", 1)
    phase = {}
    for dbg in (32, 64):                         # phase split of the one-launch form: product only / head reduction only
        ops.set_option("debug_flags", dbg)
        fn = lambda: ops.relevancy_self_chain(al, gl, B, shared_attn=shared)
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            fn()
        e1.record(); torch.cuda.synchronize()
        phase[dbg] = e0.elapsed_time(e1) / 3
    ops.set_option("debug_flags", 0)
    ops.set_option("self_chain_rows", 0)
    print("    phases of the one-launch form alone: product only %.3f ms | head reduction only %.3f ms" % (phase[32], phase[64]))
    # fp64 referee on the LAST sample (its blocks run in the last rounds of the persistent kernel)
    bl = B - 1
    R64 = torch.eye(N, dtype=torch.float64, device="cuda")
    for l in range(L):
        a_ = al[l].reshape(-1, H, N, N)[0 if shared else bl].double()
        g_ = gl[l].reshape(B, H, N, N)[bl].double()
        R64 = R64 + (a_ * g_).clamp(min=0).mean(0) @ R64
    e64 = [float((res[m][bl].double() - R64).abs().max()) for m in (0, 1)]
    print("    last sample vs fp64: two launches %.2e | one launch %.2e" % (e64[0], e64[1]))
    err = float((res[1] - res[0]).abs().max())
    top = float(res[0].abs().max())
    t0, t1 = sorted(times[0])[1], sorted(times[1])[1]
    bytes_read = L * (B * H * N * N * dt.itemsize) * (1 if shared else 2)
    flop = L * B * 2.0 * N * N * N
    print("%s\n    two launches %.3f ms | one launch %.3f ms (x%.2f) = %.0f GB/s of slabs, %.1f TFLOP/s fp32 | max |diff| %.2e (max |R| %.3g)"
          % (name, t0, t1, t0 / t1, bytes_read / t1 / 1e6, flop / t1 / 1e9, err, top), flush=True)
