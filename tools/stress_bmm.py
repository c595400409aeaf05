"""Randomised cross-check (GPU box) of the large exact-fp32 product (bmm_f32_tiles.hip) against fp64 and against the general
kernel (option bmm_tiles = 0) on random (batch, M, N, K), with / without the additive input and the NaN scrub."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer_mm_explainability_amd import ops

torch.manual_seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = 0
for case in range(cases):
    M, N = (int(torch.randint(96, 1101, ()).item()) for _ in range(2))
    K = int(torch.randint(32, 1101, ()).item())
    B = max(1, min(int(torch.randint(1, 41, ()).item()), (256 * 64 * 64 * 4) // (M * N) + 1))
    cin, nan = bool(torch.randint(0, 2, ()).item()), bool(torch.randint(0, 4, ()).item() == 0)
    a = torch.rand(B, M, K, device="cuda") / K
    b = torch.randn(B, K, N, device="cuda")
    c = torch.randn(B, M, N, device="cuda") if cin else None
    if nan:
        a[0, M // 2, K // 3] = float("nan")
    want = torch.bmm(a.double(), b.double()) + (c.double() if cin else 0)
    if nan:
        want = torch.nan_to_num(want, nan=0.0)
    ops.set_option("bmm_tiles", 1)
    got = ops.matmul(a, b, add_to=c, nan_to_zero=nan)
    ops.set_option("bmm_tiles", 0)
    old = ops.matmul(a, b, add_to=c, nan_to_zero=nan)
    ops.set_option("bmm_tiles", 1)
    scale = float(want.abs().max())
    e64, eold = float((got.double() - want).abs().max()) / scale, float((got - old).abs().max()) / scale
    ok = (e64 <= 2e-6 and eold <= 2e-6) if nan else (e64 <= 2e-6 and eold <= 2e-6 and bool(torch.isfinite(got).all()))
    bad += not ok
    print(f"case {case:3d}: B={B:2d} M={M:4d} N={N:4d} K={K:4d} cin={int(cin)} nan={int(nan)}  vs fp64 {e64:.1e}  vs general kernel {eold:.1e}  {'ok' if ok else 'FAIL'}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
