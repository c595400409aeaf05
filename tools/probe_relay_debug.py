"""Diagnostic (GPU box): which elements does the relay chain kernel pick up?  One layer, R_0 = I, so R - I = A_bar; A holds a unique
code per (head, position) and G = 1, so a wrong A_bar entry names the element that was read instead."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer_mm_explainability_amd import ops


def run(B, H, N, off, q=0):
    NN = N * N
    code = (torch.arange(H * NN, dtype=torch.float32) + 1).view(1, H, N, N).expand(B, H, N, N).contiguous()
    code = code + 100000 * torch.arange(B, dtype=torch.float32).view(B, 1, 1, 1)
    flat = torch.zeros(code.numel() + 4, device="cuda")
    flat[off:off + code.numel()] = code.reshape(-1).cuda()
    a = flat[off:off + code.numel()].view(B * H, N, N)
    g = torch.ones(B * H, N, N, device="cuda")
    ops.set_option("self_chain_algo", 3)
    ops.set_option("self_chain_relay_q", q)
    got = ops.relevancy_self_chain([a], [g], B) - torch.eye(N, device="cuda")
    ops.set_option("self_chain_algo", 0)
    ops.set_option("self_chain_relay_q", 0)
    want = code.cuda().mean(dim=1)
    bad = (got != want).nonzero()
    print(f"B={B} H={H} N={N} offset={off} q={q}: {bad.shape[0]} wrong of {B * NN}")
    if bad.shape[0]:
        # decode: with G = 1 every entry is mean_h code; report got*H - sum of the right codes of the other heads is not possible
        # in general, so print the first rows: position, wanted mean, got mean, difference x H (= displacement in elements if one head is off)
        for b, i, j in bad[:12].tolist():
            p = i * N + j
            print(f"   sample {b} position {p} (chunk {p // 4}, elem {p % 4}): want {want[b, i, j].item():.2f} got {got[b, i, j].item():.2f} "
                  f"diff*H {((got[b, i, j] - want[b, i, j]) * H).item():.1f}")
        ps = (bad[:, 1] * N + bad[:, 2])
        print("   wrong positions: min %d max %d; chunks %d..%d" % (ps.min(), ps.max(), ps.min() // 4, ps.max() // 4))


for args in [(1, 1, 33, 0), (1, 1, 33, 3), (1, 4, 33, 0), (2, 3, 97, 0), (1, 8, 77, 0), (1, 8, 77, 0, 1), (1, 2, 50, 3)]:
    run(*args)
