"""Randomised cross-check (GPU box) of the fp32 chain kernels: for random (layers, batch, heads, tokens, shared / per-sample
probabilities, R_init) the default route (layer groups by batch: relevancy_chain_groups.hip / relevancy_chain_cols.hip) and forced
group counts against the fused kernel in strict order (self_chain_algo = 1, self_chain_groups = 1): bit-identical where the route is
strict-order or compared with the fused kernel at the same number of groups, within 1e-5 otherwise."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer_mm_explainability_amd import ops

torch.manual_seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bmin, bmax = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1, 70)


def set_opts(algo, groups):
    ops.set_option("self_chain_algo", algo)
    ops.set_option("self_chain_groups", groups)


bad = 0
for case in range(cases):
    L = int(torch.randint(1, 15, ()).item())
    B = int(torch.randint(bmin, bmax + 1, ()).item())
    H = int(torch.randint(1, 13, ()).item())
    N = int(torch.randint(17, 129, ()).item())
    shared = bool(torch.randint(0, 2, ()).item())
    with_init = bool(torch.randint(0, 2, ()).item())
    attn = [torch.rand((1 if shared else B) * H, N, N, device="cuda").softmax(-1) for _ in range(L)]
    grad = [torch.randn(B * H, N, N, device="cuda") * 0.05 for _ in range(L)]
    r0 = (torch.eye(N, device="cuda") + torch.rand(B, N, N, device="cuda") * 0.1) if with_init else None
    run = lambda: ops.relevancy_self_chain(attn, grad, B, R_init=r0, shared_attn=shared).clone()   # noqa: E731
    set_opts(1, 1)
    ref = run()
    msgs = []
    set_opts(5, 0)
    if not torch.equal(run(), ref):
        msgs.append("cols != strict")
    set_opts(0, 1)
    if not torch.equal(run(), ref):
        msgs.append("auto algo with one group != strict")
    set_opts(0, 0)
    out = run()
    err = float((out - ref).abs().max())
    if not err <= 1e-5:
        msgs.append("auto: %.2e" % err)
    for g in (2, 3, 4):
        if g > L:
            continue
        set_opts(1, g)
        fused = run()
        set_opts(0, g)
        if not torch.equal(run(), fused):
            msgs.append("groups kernel != fused at G=%d" % g)
        if not float((fused - ref).abs().max()) <= 1e-5:
            msgs.append("G=%d: %.2e" % (g, float((fused - ref).abs().max())))
    set_opts(0, 0)
    status = "ok" if not msgs else "FAIL " + "; ".join(msgs)
    bad += bool(msgs)
    print(f"case {case:3d}: L={L:2d} B={B:2d} H={H:2d} N={N:3d} shared={int(shared)} init={int(with_init)} max|R|={float(ref.abs().max()):.3f} auto-vs-strict {err:.1e}  {status}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
