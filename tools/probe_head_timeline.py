"""Probe (GPU box, a build with -DMMX_HEAD_TIMELINE only): per-wave phase timeline of attn_bwd_head_kernel at the cfg-2 text shape
(B x H = 64 x 8 heads of 77 tokens, d = 64).  Every wave's lane 0 writes its s_memtime differences (shader cycles) over row 0 of its
head's dV; this script prints the distribution per phase boundary."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer_mm_explainability_amd import ops

B, H, N, D = (int(x) for x in (sys.argv[1:5] + ["64", "8", "77", "64"][len(sys.argv) - 1:]))
g = torch.Generator().manual_seed(0)
q, k, v, d_o = (torch.randn(B, N, H, D, generator=g).cuda() for _ in range(4))
probs = torch.empty(B, H, N, N, device="cuda")
dprobs = torch.empty_like(probs)
mask = torch.full((N, N), float("-inf")).triu_(1).cuda()
ops.attn_capture_fwd(q, k, v, probs, D ** -0.5, 0, mask)
names = ["staged", "barrier0", "dP tiles", "dP stored+dS", "dQ stored", "C0 barrier", "C0 done", "C1 barrier", "C1 done", "end (stores drained)"]
for rep in range(3):
    dq, dk, dv = ops.attn_capture_bwd(q, k, v, probs, d_o, dprobs, D ** -0.5, 0)
    torch.cuda.synchronize()
tl = dv[:, 0, :, :60].reshape(B * H, 5, 12).cpu()          # [head, wave, 12]
start = tl[:, :, 10]
rel = tl[:, :, :10]
print(f"{B * H} heads x 5 waves; cycles since the wave's own start (median / p10 / p90 over all waves), then per phase (median of differences)")
prev = torch.zeros_like(rel[:, :, 0])
for i, name in enumerate(names):
    x = rel[:, :, i].flatten()
    dphase = (rel[:, :, i] - prev).flatten()
    prev = rel[:, :, i]
    print(f"  {name:24s} {x.median().item():9.0f} ({x.quantile(0.1).item():7.0f} .. {x.quantile(0.9).item():7.0f})   phase {dphase.median().item():8.0f} cycles = {dphase.median().item() / 2400:6.2f} us at 2.4 GHz")
s0 = start[:, 0]
span = (s0.max() - s0.min()).item() if (s0.max() - s0.min()).item() >= 0 else float("nan")
print(f"  start-time spread of wave 0 over the heads (mod 2^24): {span:.0f} cycles")
