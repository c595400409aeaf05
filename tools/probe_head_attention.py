"""The whole-head attention-capture kernels (attention_head.hip) stand-alone at the cfg-2 shapes, next to what PLAIN library passes
reach for the same number of bytes -- the yardstick for "fraction of the HBM peak" of a kernel that moves 45-95 MB in 15-40 us.

Every timing is a hipGraph of launches that rotate over enough distinct buffer sets (> 600 MB) that nothing is served from the
256 MB Infinity Cache, HIP events around `reps` replays.

    python tools/probe_head_attention.py [path/to/libmmx_hip.so]      (another build of the library for A / B runs)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from transformer_mm_explainability_amd import _lib  # noqa: E402

if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from transformer_mm_explainability_amd import ops  # noqa: E402

DEV = "cuda"


def graph_us(launch_all, n_launches, reps=10):
    for _ in range(2):
        launch_all()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with ops.graph_capture(g):
        launch_all()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps / n_launches * 1e3


def yardsticks(mb_moved):
    """copy (half read, half written) and a flat sum (all read) moving `mb_moved` MB per launch."""
    n = int(mb_moved * 1e6 / 8) // 1024 * 1024
    sets = max(3, int(700 / mb_moved) + 1)
    src = [torch.randn(n, device=DEV) for _ in range(sets)]
    dst = [torch.empty(n, device=DEV) for _ in range(sets)]
    t_copy = graph_us(lambda: [d.copy_(s) for s, d in zip(src, dst)], sets)
    big = [torch.cat([s, d]) for s, d in zip(src, dst)]
    del src, dst
    t_sum = graph_us(lambda: [x.sum() for x in big], sets)
    rows = [x.view(-1, 1024) for x in big]
    t_rows = graph_us(lambda: [x.sum(dim=1) for x in rows], sets)
    return t_copy, t_sum, t_rows


def text_sets(B, H, N, D, sets):
    out = []
    for i in range(sets):
        g = torch.Generator(device=DEV).manual_seed(i)
        qkv = torch.randn(B, N, 3, H, D, device=DEV, generator=g)
        out.append(dict(q=qkv[:, :, 0], k=qkv[:, :, 1], v=qkv[:, :, 2], d_o=torch.randn(B, N, H, D, device=DEV, generator=g),
                        probs=torch.empty(B, H, N, N, device=DEV), grads=torch.empty(B, H, N, N, device=DEV),
                        dqkv=torch.empty(B, N, 3, H, D, device=DEV)))
    return out


def run_text(B=64, H=8, N=77, D=64):
    sets = 7
    S = text_sets(B, H, N, D, sets)
    scale = D ** -0.5
    mask = torch.full((N, N), float("-inf"), device=DEV).triu_(1)
    for s in S:
        s["o"] = ops.attn_capture_fwd(s["q"], s["k"], s["v"], s["probs"], scale, _lib.SCALE_Q_FIRST, mask)
    fwd = lambda: [ops.attn_capture_fwd(s["q"], s["k"], s["v"], s["probs"], scale, _lib.SCALE_Q_FIRST, mask) for s in S]  # noqa: E731
    t_f = graph_us(fwd, sets)
    bwd = lambda full: [ops.attn_capture_bwd(s["q"], s["k"], s["v"], s["probs"], s["d_o"], s["grads"], scale, _lib.SCALE_Q_FIRST,   # noqa: E731
                                             need_dqkv=full, out=(s["dqkv"][:, :, 0], s["dqkv"][:, :, 1], s["dqkv"][:, :, 2]) if full else None,
                                             o=s["o"]) for s in S]
    t_b = graph_us(lambda: bwd(True), sets)
    t_a = graph_us(lambda: bwd(False), sets)
    print("checksums (set 0): o %.17g  dP %.17g  dqkv %.17g" % tuple(float(S[0][k].double().sum()) for k in ("o", "grads", "dqkv")))
    vec, slab = B * H * N * D * 4, B * H * N * N * 4
    return {"fwd": (t_f, 4 * vec + slab), "bwd": (t_b, 7 * vec + 2 * slab), "bwd dP only": (t_a, 2 * vec + 2 * slab)}


def run_image(B=64, H=12, N=50, D=64):
    """shared forward: q / k / v / P of ONE image, B upstream gradients."""
    sets = 14
    S = []
    scale = D ** -0.5
    for i in range(sets):
        g = torch.Generator(device=DEV).manual_seed(100 + i)
        qkv = torch.randn(1, N, 3, H, D, device=DEV, generator=g)
        s = dict(q=qkv[:, :, 0], k=qkv[:, :, 1], v=qkv[:, :, 2], d_o=torch.randn(B, N, H, D, device=DEV, generator=g),
                 probs=torch.empty(1, H, N, N, device=DEV), grads=torch.empty(B, H, N, N, device=DEV), dqkv=torch.empty(B, N, 3, H, D, device=DEV))
        s["o"] = ops.attn_capture_fwd(s["q"], s["k"], s["v"], s["probs"], scale, _lib.SCALE_Q_FIRST, None)
        S.append(s)
    bwd = lambda full: [ops.attn_capture_bwd(s["q"], s["k"], s["v"], s["probs"], s["d_o"], s["grads"], scale, _lib.SCALE_Q_FIRST,   # noqa: E731
                                             need_dqkv=full, out=(s["dqkv"][:, :, 0], s["dqkv"][:, :, 1], s["dqkv"][:, :, 2]) if full else None,
                                             batch=B, o=s["o"]) for s in S]
    t_b = graph_us(lambda: bwd(True), sets)
    t_a = graph_us(lambda: bwd(False), sets)
    print("checksums (set 0): dP %.17g  dqkv %.17g" % tuple(float(S[0][k].double().abs().sum()) for k in ("grads", "dqkv")))
    vec, slab = B * H * N * D * 4, B * H * N * N * 4
    return {"bwd (shared forward)": (t_b, 4 * vec + slab), "bwd dP only (shared forward)": (t_a, vec + slab)}


def main():
    print("library: %s" % _lib.LIB_PATH)
    if os.environ.get("MMX_HEAD_STAGGER"):          # timing hint of attention_head.hip: later dispatch rounds start N x 64 cycles late
        ops.set_option("attn_head", 1 | (int(os.environ["MMX_HEAD_STAGGER"]) << 16))
        print("stagger units: %s" % os.environ["MMX_HEAD_STAGGER"])
    for name, res in (("text  B=64 H=8  N=77 d=64", run_text()), ("image B=64 H=12 N=50 d=64", run_image())):
        for leg, (t, nbytes) in res.items():
            print("%s  %-30s %6.1f us  %6.1f MB  %.2f TB/s = %4.1f %% of 8 TB/s" % (name, leg, t, nbytes / 1e6, nbytes / t / 1e6, nbytes / t / 8e4))
    if len(sys.argv) <= 2 and not os.environ.get("MMX_HEAD_STAGGER"):
        for mb in (45, 95):
            t_copy, t_sum, t_rows = yardsticks(mb)
            print("library passes moving %3d MB per launch: copy_ %5.1f us = %.2f TB/s | flat sum %5.1f us = %.2f TB/s | row sums [n, 1024] %5.1f us = %.2f TB/s"
                  % (mb, t_copy, mb / t_copy, t_sum, mb / t_sum, t_rows, mb / t_rows))


if __name__ == "__main__":
    main()
