"""Library fp32 GEMM layouts for the input-gradient products dX = dY . W (W = nn.Linear.weight [out, in]): the NN form
``matmul(dY, W)`` against the NT form ``F.linear(dY, W^T contiguous)`` and the accumulating ``addmm_`` of both -- which kernel the
hipBLASLt heuristic picks differs per layout.  Shapes: DETR K = 10 / 20 (M = 9500 / 19000), LXMERT B = 32 (M = 448 / 1152)."""

import torch
import torch.nn.functional as F


def t(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


shapes = [(9500, 256, 2048), (9500, 2048, 256), (9500, 256, 256), (9500, 768, 256), (9500, 512, 256), (19000, 256, 2048),
          (19000, 2048, 256), (448, 768, 768), (448, 3072, 768), (448, 768, 3072), (1152, 768, 768), (1152, 3072, 768),
          (1152, 768, 3072), (448, 2304, 768), (1152, 2304, 768)]
print("%-22s %10s %10s %10s %10s   TFLOP/s(best)" % ("M, K(out), N(in)", "NN us", "NT us", "NN addmm_", "NT addmm_"))
for M, Kd, N in shapes:
    dY = torch.randn(M, Kd, device="cuda")
    W = torch.randn(Kd, N, device="cuda") * 0.02          # nn.Linear.weight [out = Kd, in = N]
    Wt = W.t().contiguous()                               # [N, Kd]
    acc = torch.randn(M, N, device="cuda")
    r = [t(lambda: torch.matmul(dY, W)), t(lambda: F.linear(dY, Wt)), t(lambda: acc.addmm_(dY, W)),
         t(lambda: acc.addmm_(dY, Wt.t()))]
    print("%-22s %10.1f %10.1f %10.1f %10.1f   %.1f" % ((M, Kd, N), *r, 2.0 * M * Kd * N / min(r[:2]) / 1e6))
