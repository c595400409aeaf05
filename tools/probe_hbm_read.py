"""What a plain streaming pass reaches on this box (context for the chain kernel's roofline fraction: its peak column is the 8 TB/s
datasheet number): library reductions / copies over buffers of the chain's size (291 MB) and larger, timed with HIP events."""
import torch


def us(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for mb in (291, 1024, 4096):
    n = (mb * 1000 * 1000 // 4) // 1024 * 1024
    x = torch.randn(n, device="cuda")
    y = torch.empty_like(x)
    x4 = x.view(-1, 4)
    t_sum, t_max, t_copy, t_mul = us(lambda: x.sum()), us(lambda: x.amax()), us(lambda: y.copy_(x)), us(lambda: torch.mul(x, 2.0, out=y))
    rows = x.view(-1, 1024)
    t_rows = us(lambda: rows.sum(dim=1))
    print("%5d MB: sum %.1f us = %.2f TB/s | amax %.2f TB/s | row sums [n, 1024] %.2f TB/s | copy (r+w) %.2f TB/s | mul out= (r+w) %.2f TB/s"
          % (mb, t_sum, n * 4 / t_sum / 1e6, n * 4 / t_max / 1e6, n * 4 / t_rows / 1e6, 2 * n * 4 / t_copy / 1e6, 2 * n * 4 / t_mul / 1e6))
