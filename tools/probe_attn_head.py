"""Probe (GPU box): attention-capture forward / backward at the cfg-2 shapes, register-resident ("head") vs whole-head-in-LDS
("small") kernels, hipGraph-replay timed; prints algorithmic GB/s (q,k,v,dO in; P / dP, O / dq,dk,dv out) and exact-fp32
MFMA TFLOP/s next to the time, and the shared-forward (batch-stride-0) backward the image tower really runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer_mm_explainability_amd import ops


def bench(fn, reps=10, iters=10, warm=2):
    """us per call: `reps` back-to-back calls captured into ONE hipGraph (a one-kernel graph would time the ~10 us replay
    floor, not the kernel), replayed `iters` times."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(reps):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        graph.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (iters * reps) * 1e3


def main():
    for (B, H, N, D, name) in [(64, 12, 50, 64, "img"), (64, 8, 77, 64, "txt"), (32, 12, 112, 64, "bert112"), (128, 12, 20, 64, "lx20")]:
        qkv = torch.randn(B, N, 3, H, D, device="cuda")
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
        probs = torch.empty(B, H, N, N, device="cuda"); dprobs = torch.empty_like(probs)
        d_o = torch.randn(B, N, H, D, device="cuda")
        dqkv = torch.empty_like(qkv)
        out = (dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2])
        mask = torch.full((N, N), float("-inf"), device="cuda").triu_(1) if name == "txt" else None
        slab = B * H * N * N * 4
        qkvb = B * H * N * D * 4
        fwd_bytes, bwd_bytes = 4 * qkvb + slab, 7 * qkvb + 2 * slab
        fwd_flop, bwd_flop = 4 * B * H * N * N * D, 8 * B * H * N * N * D
        for head in (1, 0):
            ops.set_option("attn_head", head)
            tag = "head " if head else "small"
            us = bench(lambda: ops.attn_capture_fwd(q, k, v, probs, D ** -0.5, 0, mask))
            print(f"{name}: fwd {tag}: {us:6.1f} us  {fwd_bytes / us / 1e3:7.0f} GB/s  {fwd_flop / us / 1e6:6.1f} TF/s")
            us = bench(lambda: ops.attn_capture_bwd(q, k, v, probs, d_o, dprobs, D ** -0.5, 0, out=out))
            print(f"{name}: bwd {tag}: {us:6.1f} us  {bwd_bytes / us / 1e3:7.0f} GB/s  {bwd_flop / us / 1e6:6.1f} TF/s")
            us = bench(lambda: ops.attn_capture_bwd(q, k, v, probs, d_o, dprobs, D ** -0.5, 0, need_dqkv=False))
            print(f"{name}: bwd {tag} (dP only): {us:6.1f} us")
        if name == "img":   # shared-forward backward: q/k/v/P of ONE sample, B upstream gradients
            q1, k1, v1 = (t[:1] for t in (q, k, v))
            p1 = probs[:1].contiguous()
            for head in (1, 0):
                ops.set_option("attn_head", head)
                us = bench(lambda: ops.attn_capture_bwd(q1, k1, v1, p1, d_o, dprobs, D ** -0.5, 0, out=out, batch=B))
                print(f"{name}: bwd shared-forward {'head ' if head else 'small'}: {us:6.1f} us  {(4 * qkvb + slab) / us / 1e3:7.0f} GB/s")
    ops.set_option("attn_head", 1)


if __name__ == "__main__":
    main()
