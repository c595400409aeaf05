"""-m gpu: the HIP kernels (through the C-ABI) against the numpy oracle on seeded inputs.

Tolerance: 1e-5 absolute on relevancy values (BASELINE.json north_star: "within 1e-5 (fp32)"); both sides
are fp32 and differ only in summation order.
"""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import relevancy_np as onp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import parity  # noqa: E402

pytestmark = pytest.mark.gpu
ATOL = 1e-5


@pytest.fixture(scope="module")
def ops():
    from transformer_mm_explainability_amd import ops as _ops
    return _ops


def dev(x):
    return torch.as_tensor(np.asarray(x)).cuda()


def make_layers(seed, L, B, H, N, causal=False, gscale=0.05):
    g = torch.Generator().manual_seed(seed)
    attn, grad = [], []
    for _ in range(L):
        s = torch.randn(B * H, N, N, generator=g)
        if causal:
            s = s + torch.full((N, N), float("-inf")).triu_(1)
        attn.append(s.softmax(-1))
        grad.append(torch.randn(B * H, N, N, generator=g) * gscale)
    return attn, grad


def close(a, b, atol=ATOL, relmax=parity.RELMAX):
    """``tests/parity.close`` with this suite's elementwise bound (``atol + 1e-5 |ref|``) PLUS the suite-wide relative bound on the
    largest error (``relmax`` x max |ref|) and the error record (``profiles/rNN_parity.json``); the record key carries the call
    site's line and the compared shape, so every distinct comparison of this file is in it."""
    line = sys._getframe(1).f_lineno
    shape = "x".join(str(d) for d in np.shape(b)) or "scalar"
    parity.close(a, b, atol=atol, rtol=1e-5, what="L%d[%s]" % (line, shape), relmax=relmax, rel_always=True)


def test_mfma_matmul_asymmetric(ops):
    """A = I check with an asymmetric B, then random rectangular shapes incl. transposed A and accumulate."""
    B = torch.arange(16 * 24, dtype=torch.float32).reshape(16, 24) * 0.01
    close(ops.matmul(torch.eye(16).cuda(), B.cuda()), B.numpy())
    g = torch.Generator().manual_seed(0)
    for (m, n, k) in [(5, 7, 3), (64, 64, 64), (100, 950, 100), (77, 77, 77), (130, 65, 200)]:
        a = torch.randn(m, k, generator=g)
        b = torch.randn(k, n, generator=g)
        c = torch.randn(m, n, generator=g)
        close(ops.matmul(a.cuda(), b.cuda()), a.numpy() @ b.numpy(), atol=1e-4)
        close(ops.matmul(a.t().contiguous().cuda(), b.cuda(), trans_a=True), a.numpy() @ b.numpy(), atol=1e-4)
        close(ops.matmul(a.cuda(), b.cuda(), add_to=c.cuda()), c.numpy() + a.numpy() @ b.numpy(), atol=1e-4)
    a = torch.randn(3, 20, 30, generator=g)
    b = torch.randn(3, 30, 10, generator=g)
    close(ops.matmul(a.cuda(), b.cuda()), np.matmul(a.numpy(), b.numpy()), atol=1e-4)


@pytest.mark.parametrize("B,H,Nq,Nk", [(1, 12, 9, 13), (4, 8, 77, 77), (2, 12, 50, 50), (1, 3, 1, 5), (3, 1, 33, 2)])
def test_avg_heads(ops, B, H, Nq, Nk):
    g = torch.Generator().manual_seed(B * 1000 + Nq)
    cam = torch.randn(B * H, Nq, Nk, generator=g).softmax(-1)
    grad = torch.randn(B * H, Nq, Nk, generator=g)
    want = onp.avg_heads_batched(cam.numpy(), grad.numpy(), B)
    close(ops.avg_heads(cam.cuda(), grad.cuda(), batch_size=B), want, atol=1e-6)


def test_avg_heads_nan_and_lowp(ops):
    cam = torch.rand(4, 6, 6)
    grad = torch.randn(4, 6, 6)
    grad[1, 2, 3] = float("nan")
    close(ops.avg_heads(cam.cuda(), grad.cuda()), onp.avg_heads(cam.numpy(), grad.numpy())[None], atol=1e-6)
    for dt in (torch.bfloat16, torch.float16):
        c, g = torch.rand(8, 10, 10).to(dt), torch.randn(8, 10, 10).to(dt)
        want = onp.avg_heads(c.float().numpy(), g.float().numpy())[None]
        close(ops.avg_heads(c.cuda(), g.cuda()), want, atol=1e-6)


@pytest.mark.parametrize("L,B,H,N,causal", [
    (12, 4, 12, 50, False),   # CLIP ViT-B/32 image tower shape
    (12, 3, 8, 77, True),     # CLIP text tower shape (causal)
    (3, 2, 2, 5, False), (1, 1, 1, 1, False), (2, 5, 3, 16, False), (4, 2, 4, 17, False),
    (6, 2, 8, 100, False), (2, 1, 4, 128, False), (5, 3, 12, 36, False), (0, 2, 1, 7, False),
])
# 0: auto (layer groups by batch; fp32 slabs: relevancy_chain_groups.hip / relevancy_chain_cols.hip), 1: the fused kernel everywhere,
# 5: relevancy_chain_cols.hip wherever it applies
@pytest.mark.parametrize("algo", [0, 1, 5])
def test_self_chain_fused(chain_options, L, B, H, N, causal, algo):
    ops = chain_options
    ops.set_option("self_chain_algo", algo)
    attn, grad = make_layers(L * 100 + N, max(L, 1), B, H, N, causal)
    attn, grad = attn[:L], grad[:L]
    want = onp.self_chain([a.numpy() for a in attn], [g.numpy() for g in grad], B) if L else \
        np.broadcast_to(np.eye(N, dtype=np.float32), (B, N, N))
    if L == 0:
        got = ops.relevancy_self_chain([], [], B, R_init=torch.eye(N).cuda())
    else:
        got = ops.relevancy_self_chain([a.cuda() for a in attn], [g.cuda() for g in grad], B)
    close(got, want)


@pytest.mark.parametrize("slab_dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("L,B,H,N,causal,shared", [
    (12, 4, 12, 50, False, False), (12, 3, 8, 77, True, False), (3, 3, 2, 12, True, False), (3, 2, 2, 17, False, True),
    (4, 2, 4, 130, False, False), (2, 2, 3, 197, False, True),
])
def test_self_chain_half_vs_torch_half_chain(ops, L, B, H, N, causal, shared, slab_dtype):
    """``mmx_relevancy_self_chain_half`` (ChainPlan(half_chain=True)): the reference's fp16 chain -- R created in the dtype of the
    fp16 probabilities, notebook cell 6:20,43 -- against the same torch ops on fp16 CPU tensors (oracle/clip_torch.chain_half).
    Both round every tensor-level result to fp16 and sum in fp32; only the ORDER of the fp32 sums differs (MFMA tiles vs the CPU
    GEMM), so almost every element is equal and the rest are within two fp16 places of the largest entry.  N <= 128: the fused
    kernel; larger: avg_heads + bmm + the rounding add."""
    from oracle import clip_torch
    attn, grad = make_layers(L * 7 + N, L, B, H, N, causal)
    attn = [a.to(torch.float16) for a in attn]
    grad = [(g * 10).to(torch.float16) for g in grad]
    if shared:                                         # one forward shared by the batch: the probabilities of sample 0
        attn = [a.view(B, H, N, N)[:1].expand(B, H, N, N).reshape(B * H, N, N).contiguous() for a in attn]
    want = clip_torch.chain_half(attn, grad, B).float().numpy()
    ca = [(a.view(B, H, N, N)[0] if shared else a).to(slab_dtype).contiguous().cuda() for a in attn]
    cg = [g.to(slab_dtype).cuda() for g in grad]
    plan = ops.ChainPlan(ca, cg, B, shared_attn=shared, half_chain=True)
    for _ in range(2):
        got = plan.launch()
        assert got.dtype == torch.float16
        got = got.float().cpu().numpy()
        diff = np.abs(got - want)
        assert diff.max() <= 2.0 ** -9 * np.abs(want).max(), (diff.max(), np.abs(want).max())
        assert (diff == 0).mean() > 0.98, (diff == 0).mean()
    with pytest.raises(ops.MMXError):
        ops.ChainPlan([a.to(torch.bfloat16) for a in ca], [g.to(torch.bfloat16) for g in cg], B, shared_attn=shared, half_chain=True)


@pytest.mark.parametrize("algo", [1, 4])        # 1: self_chain_fused_kernel, 4 (= auto for fp32 slabs): self_chain_groups_kernel
@pytest.mark.parametrize("groups", [2, 3, 4])
@pytest.mark.parametrize("L,B,H,N,causal,with_init,shared", [
    (12, 4, 12, 50, False, False, False), (12, 3, 8, 77, True, True, False), (5, 2, 4, 33, False, True, False),
    (2, 2, 2, 128, False, False, False), (12, 2, 3, 100, False, False, True), (7, 5, 5, 17, False, True, True),
    (4, 1, 9, 91, False, False, False), (6, 11, 1, 128, False, True, False), (8, 3, 16, 16, False, False, False),
])
def test_self_chain_layer_groups(chain_options, algo, groups, L, B, H, N, causal, with_init, shared):
    """Layer-group split: partial products re-associated at the group boundaries, combined by the last arriver.
    Same 1e-5 bar against the sequential oracle; repeated launches reuse scratch and counters.  The two kernels that implement
    it differ in how the slabs reach LDS, not in the arithmetic: their results are equal bit for bit."""
    ops = chain_options
    attn, grad = make_layers(L * 10 + N, L, 1 if shared else B, H, N, causal)
    if shared:
        grad = [torch.randn(B * H, N, N) * 0.05 for _ in range(L)]
    R0 = (torch.eye(N) + torch.rand(B, N, N) * 0.1) if with_init else None
    want = np.broadcast_to(np.eye(N, dtype=np.float32), (B, N, N)).copy() if R0 is None else R0.numpy().copy()
    for a, g in zip(attn, grad):
        full = a.reshape(1, H, N, N).expand(B, H, N, N).reshape(B * H, N, N) if shared else a
        cam = onp.avg_heads_batched(full.numpy(), g.numpy(), B)
        want = want + np.matmul(cam, want)
    ca, cg = [a.cuda() for a in attn], [g.cuda() for g in grad]
    ops.set_option("self_chain_groups", groups)
    outs = {}
    for which in sorted({1, algo}):
        ops.set_option("self_chain_algo", which)
        for _ in range(3):
            got = ops.relevancy_self_chain(ca, cg, B, R_init=R0.cuda() if with_init else None, shared_attn=shared)
            close(got, want)
        outs[which] = got.clone()
    assert torch.equal(outs[1], outs[algo])


def test_self_chain_auto_groups(chain_options):
    """The automatic number of layer groups, read off the scratch the C-ABI asks for (B tickets, 256-byte aligned, + B x G partial
    products): the fewest groups that put a workgroup on ~70 % of the CUs, at most 4, never more workgroups than CUs; one group
    (no scratch) below 1 MB per sample and when the batch alone fills the chip.  On the 256 CUs of an MI355X."""
    from transformer_mm_explainability_amd import _lib
    lib = _lib.lib()
    cus = torch.cuda.get_device_properties(0).multi_processor_count

    def groups(L, B, H, N):
        need = lib.mmx_self_chain_workspace_bytes(L, B, H, N, 0, _lib.MMX_F32)
        if need == 0:
            return 1
        tickets = (4 * B + 255) // 256 * 256
        per_group = (4 * B * N * N + 255) // 256 * 256
        g, rem = divmod(need - tickets, 4 * B * N * N)
        assert need - tickets == (4 * B * g * N * N + 255) // 256 * 256 or rem < 256, (need, tickets, per_group)
        return g
    if cus != 256:
        pytest.skip("rule pinned for 256 CUs")
    assert [groups(12, B, 8, 77) for B in (1, 16, 32, 59, 60, 64, 85, 86, 96, 128, 129, 160, 256)] == \
        [4, 4, 4, 4, 3, 3, 3, 2, 2, 2, 1, 1, 1]
    assert groups(12, 64, 12, 50) == 3 and groups(1, 64, 8, 77) == 1 and groups(3, 2, 2, 12) == 1 and groups(2, 64, 8, 77) == 1 \
        and groups(4, 64, 8, 77) == 3 and groups(3, 16, 8, 77) == 3


@pytest.mark.parametrize("L,B,H,N,groups", [(12, 8, 8, 77, 0), (12, 4, 12, 50, 0), (3, 2, 5, 44, 1), (2, 1, 16, 128, 0), (5, 3, 8, 20, 2),
                                            (4, 2, 3, 64, 0), (1, 1, 1, 9, 1), (3, 2, 7, 112, 0), (2, 2, 6, 96, 1)])
def test_self_chain_pipelined_stream_waves_bit_identical(ops, L, B, H, N, groups):
    """Option ``self_chain_pipe`` (the stream waves of the fused chain as a software pipeline of raw buffer loads: the next batch
    in flight across the reduction, the LDS write and the per-layer barrier) changes how the slabs are fetched, not the
    arithmetic: heads are summed in the same order, the results equal the plain loop bit for bit (odd head counts, ragged last
    chunks, one / several layer groups)."""
    attn, grad = make_layers(31 + H + N, L, B, H, N)
    outs = []
    try:
        ops.set_option("self_chain_algo", 1)     # (the option belongs to self_chain_fused_kernel)
        ops.set_option("self_chain_groups", groups)
        for pipe in (4, 2, 1, 0):                 # up to 4 / 2 / 1 contiguous chunks per lane and head; 0: the plain loop
            ops.set_option("self_chain_pipe", pipe)
            outs.append(ops.relevancy_self_chain([dev(a) for a in attn], [dev(g) for g in grad], B))
    finally:
        ops.set_option("self_chain_pipe", 4)
        ops.set_option("self_chain_groups", 0)
        ops.set_option("self_chain_algo", 0)
    assert all(torch.equal(o, outs[-1]) for o in outs[:-1])


@pytest.mark.parametrize("B,M,N,K,cin,nan", [
    (3, 577, 577, 577, True, False),     # ViT-L/14@336 chain product: ragged edges in all three dimensions
    (32, 197, 197, 197, True, False),    # ViT-B/16 (at batch 4 the grid is below one workgroup per CU: general kernel)
    (4, 197, 197, 197, True, False),
    (2, 950, 950, 950, True, False),     # DETR encoder
    (40, 100, 950, 950, False, True),    # rule 10 at DETR size, batched, with the NaN scrub
    (300, 130, 100, 64, False, False),   # small tiles, many batch entries
    (2, 1050, 1050, 1050, True, False),
])
def test_bmm_tiles_kernel(chain_options, B, M, N, K, cin, nan):
    """``bmm_f32_tiles.hip`` (32 x 32 x 2 exact-fp32 MFMA, b128 operand reads, one barrier per K slab) against fp64 and against the
    general kernel it replaces for the large plain products (option ``bmm_tiles`` = 0): same exact-fp32 products, another
    summation order."""
    ops = chain_options
    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.rand(B, M, K, generator=g) / K).cuda()
    b = torch.randn(B, K, N, generator=g).cuda()
    c = torch.randn(B, M, N, generator=g).cuda() if cin else None
    if nan:
        a[0, 3, 5] = float("nan")
    want = torch.bmm(a.double(), b.double()) + (c.double() if cin else 0)
    if nan:
        want = torch.nan_to_num(want, nan=0.0)
    got = ops.matmul(a, b, add_to=c, nan_to_zero=nan)
    ops.set_option("bmm_tiles", 0)
    old = ops.matmul(a, b, add_to=c, nan_to_zero=nan)
    scale = float(want.abs().max())
    assert float((got.double() - want).abs().max()) <= 2e-6 * scale
    assert float((got - old).abs().max()) <= 2e-6 * scale


@pytest.fixture
def attn_options(ops):
    """Options of the attention kernels are process-global: whatever a test sets is put back, also when it fails."""
    yield ops
    for key, value in (("attn_head", 1), ("attn_stream", 1), ("attn_fwd_split", 1), ("attn_bf16_v3", 3), ("attn_head_tile_skip", 1)):
        ops.set_option(key, value)


@pytest.fixture
def chain_options(ops):
    """Options of the chain kernels are process-global: whatever a test sets is put back, also when it fails."""
    yield ops
    for key, value in (("self_chain_algo", 0), ("self_chain_groups", 0), ("self_chain_pipe", 4), ("self_chain_nt", 1),
                       ("self_chain_cols_c", 0), ("self_chain_cols_nb", 0), ("self_chain_rows", 0),
                       ("debug_flags", 0), ("bmm_tiles", 1)):
        ops.set_option(key, value)


@pytest.mark.parametrize("L,B,H,N,causal,with_init,shared,offset", [
    (12, 64, 8, 77, True, False, False, 0),     # cfg-2 text tower at the bench's batch, odd N^2
    (12, 64, 12, 50, False, False, True, 0),    # cfg-2 image tower, one forward shared by the batch
    (12, 5, 8, 77, True, True, False, 3),       # batch not a multiple of 8, R_init, slabs 12 bytes above a 16-byte boundary
    (3, 2, 2, 12, False, False, False, 1), (2, 3, 4, 128, False, True, False, 0), (4, 9, 5, 33, False, False, False, 2),
    (6, 1, 12, 112, False, False, False, 0),    # VisualBERT's shape: 7 column slabs, three ring images
    (2, 2, 3, 97, False, False, False, 0),
    (1, 2, 1, 7, False, False, False, 1), (3, 130, 2, 40, False, False, False, 0),
])
def test_self_chain_strict_order_bit_identical(chain_options, L, B, H, N, causal, with_init, shared, offset):
    """K1c (``relevancy_chain_cols.hip``: barrier-free stream waves feeding a ring of A_bar images, strict layer order, optionally
    the columns of R split over several workgroups per sample) == the fused kernel with one group BIT FOR BIT (same head order,
    same MFMA chain) and within 1e-5 of the oracle; workgroups per sample / ring depth change the schedule, not the bits."""
    ops = chain_options
    attn, grad = make_layers(L * 13 + N + B, L, B, H, N, causal)
    if shared:
        attn = [a.view(B, H, N, N)[:1].expand(B, H, N, N).reshape(B * H, N, N).contiguous() for a in attn]
    R0 = (torch.eye(N) + torch.rand(B, N, N) * 0.1) if with_init else None
    want = np.broadcast_to(np.eye(N, dtype=np.float32), (B, N, N)).copy() if R0 is None else R0.numpy().copy()
    for a, g in zip(attn, grad):
        want = want + np.matmul(onp.avg_heads_batched(a.numpy(), g.numpy(), B), want)

    def place(t):                                # a contiguous view `offset` elements above an aligned allocation
        flat = torch.empty(t.numel() + 4, device="cuda")
        flat[offset:offset + t.numel()] = t.reshape(-1).cuda()
        return flat[offset:offset + t.numel()].view(t.shape)
    ca = [place(a.view(B, H, N, N)[0] if shared else a) for a in attn]
    cg = [place(g) for g in grad]
    r0 = R0.cuda() if with_init else None
    run = lambda: ops.relevancy_self_chain(ca, cg, B, R_init=r0, shared_attn=shared).clone()   # noqa: E731
    ops.set_option("self_chain_algo", 1)
    ops.set_option("self_chain_groups", 1)
    ref = run()
    close(ref, want)
    # K1c (``relevancy_chain_cols.hip``): the columns of R split over the workgroups of a sample, every workgroup reduces the full
    # A_bar -- the same sequential chain; workgroups per sample / ring depth change the schedule, not the bits
    ops.set_option("self_chain_algo", 5)
    for c, nb in ((0, 0), (0, 0), (1, 0), (2, 1), (3, 2), (8, 3)):
        ops.set_option("self_chain_cols_c", c)
        ops.set_option("self_chain_cols_nb", nb)
        got = run()
        assert torch.equal(got, ref), ("cols", c, nb, float((got - ref).abs().max()))


@pytest.mark.parametrize("N,M,L", [(197, 0, 3), (130, 0, 3), (20, 36, 3), (100, 300, 3), (577, 0, 5), (197, 64, 4)])
def test_self_chain_split_and_rule7(chain_options, N, M, L):
    """N > 128 (or a second right-hand side): per layer avg_heads + exact-fp32 product (bmm_f32_tiles.hip for the large ones)."""
    ops = chain_options
    B, H = 2, 4
    attn, grad = make_layers(N + M, L, B, H, N)
    g = torch.Generator().manual_seed(7)
    R0 = torch.eye(N).expand(B, N, N).contiguous()
    sq0 = torch.rand(B, N, M, generator=g) * 0.1 if M else None
    R, SQ = R0.numpy().copy(), (sq0.numpy().copy() if M else None)
    for a, gr in zip(attn, grad):
        cam = onp.avg_heads_batched(a.numpy(), gr.numpy(), B)
        if M:
            SQ = SQ + np.matmul(cam, SQ)
        R = R + np.matmul(cam, R)
    got = ops.relevancy_self_chain([a.cuda() for a in attn], [x.cuda() for x in grad], B,
                                   R_sq_init=sq0.cuda() if M else None)
    if M:
        close(got[0], R)
        close(got[1], SQ)
    else:
        close(got, R)

def test_self_chain_bf16_capture(ops):
    L, B, H, N = 4, 2, 4, 50
    attn, grad = make_layers(5, L, B, H, N)
    a16 = [a.bfloat16() for a in attn]
    g16 = [g.bfloat16() for g in grad]
    want = onp.self_chain([a.float().numpy() for a in a16], [g.float().numpy() for g in g16], B)
    close(ops.relevancy_self_chain([a.cuda() for a in a16], [g.cuda() for g in g16], B), want)


def test_rules_golden(ops, golden):
    g = golden("rules")
    close(ops.handle_residual(dev(g["R_ss"])), g["handle_residual_ss"])
    with pytest.raises(AssertionError):
        ops.handle_residual(dev(g["R_ss"] - 2 * np.eye(g["R_ss"].shape[0], dtype=np.float32)))
    close(ops.mm_attention_rules(dev(g["R_ss"]), dev(g["R_qq"]), dev(g["cam_sq"]), nan_to_zero=True), g["mm_detr_norm"])
    close(ops.mm_attention_rules(dev(g["R_ss"]), dev(g["R_qq"]), dev(g["cam_sq"]), apply_normalization=False,
                                 nan_to_zero=True), g["mm_detr_nonorm"])
    close(ops.mm_attention_rules(dev(g["R_ss"]), dev(g["R_qq"]), dev(g["cam_sq"]), apply_self_in_rule_10=False,
                                 nan_to_zero=True), g["mm_detr_noself"])
    ns, nq = g["cam_sq"].shape
    eye_s, eye_q = torch.eye(ns).cuda(), torch.eye(nq).cuda()
    close(ops.mm_attention_rules(eye_s, eye_q, dev(g["cam_sq"]), nan_to_zero=True), g["mm_detr_nan"])
    sq, ss = ops.mm_attention_rules(dev(g["R_ss"]), dev(g["R_qq"]), dev(g["cam_sq"]), R_qs=dev(g["R_qs"]))
    close(sq, g["mm_lx_sq_add"])
    close(ss, g["mm_lx_ss_add"])
    sq, ss = ops.mm_attention_rules(eye_s, eye_q, dev(g["cam_sq"]), R_qs=dev(g["R_qs"]))
    close(sq, g["mm_lx_nan_sq_add"])
    close(ss, g["mm_lx_nan_ss_add"])
    # apply_self_in_rule_10=False returns cam_sq -- the DETR flavour scrubs ITS NaNs as well (reference
    # DETR/...:40-42), and handle_residual (with its diag >= 0 assert) still runs when normalisation is on (:36-38)
    cam_nan = dev(g["cam_sq"]).clone()
    cam_nan[1, 2] = float("nan")
    out = ops.mm_attention_rules(dev(g["R_ss"]), dev(g["R_qq"]), cam_nan, apply_self_in_rule_10=False, nan_to_zero=True)
    want = g["cam_sq"].copy()
    want[1, 2] = 0
    close(out, want)
    kept = ops.mm_attention_rules(dev(g["R_ss"]), dev(g["R_qq"]), cam_nan, apply_self_in_rule_10=False)
    assert torch.isnan(kept[1, 2]) and torch.isnan(kept).sum() == 1          # LXMERT flavour: NaN propagates
    bad = dev(g["R_ss"] - 2 * np.eye(g["R_ss"].shape[0], dtype=np.float32))
    with pytest.raises(AssertionError):
        ops.mm_attention_rules(bad, dev(g["R_qq"]), dev(g["cam_sq"]), apply_self_in_rule_10=False, nan_to_zero=True)
    ops.mm_attention_rules(bad, dev(g["R_qq"]), dev(g["cam_sq"]), apply_normalization=False,
                           apply_self_in_rule_10=False, nan_to_zero=True)   # no normalisation -> no assert (reference)


def test_rollout_golden(ops, golden):
    g = golden("rules")
    mats = [dev(m) for m in g["rollout_in"]]
    close(ops.rollout_chain(mats, True), g["rollout_detr"])
    close(ops.rollout_chain(mats[2:], True), g["rollout_detr_s2"])
    matsb = [dev(m) for m in g["rollout_vb_in"]]
    close(ops.rollout_chain(matsb, False), g["rollout_vb"])
    close(ops.rollout_chain(matsb[1:], False), g["rollout_vb_s1"])


def torch_attention(q, k, v, scale, mode, mask):
    """Reference of the op in float64: q,k,v [B,H,N,D]."""
    q, k, v = q.double(), k.double(), v.double()
    s = (q * scale) @ k.transpose(-1, -2) if mode == 0 else (q @ k.transpose(-1, -2)) / scale
    if mask is not None:
        s = s + mask.double()
    p = s.softmax(-1)
    return p, p @ v


@pytest.mark.parametrize("B,H,Nq,Nk,D,mode,masked", [
    (2, 12, 50, 50, 64, 0, False), (2, 8, 77, 77, 64, 0, True), (1, 8, 100, 180, 32, 0, False),
    (2, 12, 14, 36, 64, 1, False), (1, 2, 1, 3, 16, 1, False), (1, 4, 197, 197, 64, 0, False),
    (1, 2, 33, 130, 48, 1, True), (1, 4, 15, 15, 8, 0, False), (1, 4, 7, 15, 8, 0, False), (2, 3, 9, 20, 4, 1, False),
    (1, 2, 20, 20, 24, 0, False), (1, 8, 200, 330, 32, 0, False), (2, 4, 130, 70, 64, 1, True),
    (1, 3, 65, 129, 48, 0, True), (2, 12, 128, 128, 64, 1, True), (1, 4, 14, 36, 64, 1, False), (1, 2, 36, 14, 64, 1, False),
    (1, 2, 250, 100, 32, 0, False), (3, 2, 77, 77, 20, 0, True),
    (2, 3, 21, 40, 6, 0, True), (1, 2, 70, 70, 10, 1, False),       # head_dim % 4 != 0: only the general tiled kernels take these
])
@pytest.mark.parametrize("path", ["head", "stream", "tiled"])
def test_attn_capture_fwd_bwd(attn_options, B, H, Nq, Nk, D, mode, masked, path):
    """head: register-resident whole-head kernels where eligible (else the next path); stream: the long-sequence kernels
    (K/V streamed in 64-row tiles) for every shape they accept; tiled: the general kernels (any head_dim, any alignment) for
    every shape -- the default for what the other two turn down."""
    ops = attn_options
    ops.set_option("attn_head", int(path == "head"))
    ops.set_option("attn_stream", int(path != "tiled"))
    g = torch.Generator().manual_seed(Nq * 7 + Nk)
    q = torch.randn(B, Nq, H, D, generator=g)
    k = torch.randn(B, Nk, H, D, generator=g)
    v = torch.randn(B, Nk, H, D, generator=g)
    d_o = torch.randn(B, Nq, H, D, generator=g)
    scale = D ** -0.5 if mode == 0 else D ** 0.5
    mask = None
    if masked:
        mask = torch.full((Nq, Nk), float("-inf")).triu_(1) if Nq == Nk else \
            (torch.randn(B, 1, Nk, generator=g) > 1.0).float() * -10000.0
    qr, kr, vr = (t.permute(0, 2, 1, 3).clone().requires_grad_(True) for t in (q, k, v))
    mref = mask if mask is None or mask.dim() == 2 else mask[:, None]
    p_ref, o_ref = torch_attention(qr, kr, vr, scale, mode, mref)
    p_ref.retain_grad()
    (o_ref * d_o.permute(0, 2, 1, 3).double()).sum().backward()

    probs = torch.empty(B, H, Nq, Nk, device="cuda")
    dprobs = torch.empty(B, H, Nq, Nk, device="cuda")
    qc, kc, vc = q.cuda(), k.cuda(), v.cuda()
    o = ops.attn_capture_fwd(qc, kc, vc, probs, scale, mode, mask.cuda() if mask is not None else None)
    close(probs, p_ref.detach().float().numpy(), atol=2e-6)
    close(o.permute(0, 2, 1, 3), o_ref.detach().float().numpy(), atol=1e-5)
    dq, dk, dv = ops.attn_capture_bwd(qc, kc, vc, probs, d_o.cuda(), dprobs, scale, mode)
    close(dprobs, p_ref.grad.float().numpy(), atol=2e-5)
    close(dq.permute(0, 2, 1, 3), qr.grad.float().numpy(), atol=2e-5)
    close(dk.permute(0, 2, 1, 3), kr.grad.float().numpy(), atol=2e-5)
    close(dv.permute(0, 2, 1, 3), vr.grad.float().numpy(), atol=2e-5)
    # with the forward's O the streaming backward takes delta = rowsum(dO * O) instead of sweeping the keys
    dq2, dk2, dv2 = ops.attn_capture_bwd(qc, kc, vc, probs, d_o.cuda(), torch.empty_like(dprobs), scale, mode, o=o)
    close(dq2.permute(0, 2, 1, 3), qr.grad.float().numpy(), atol=2e-5)
    close(dk2.permute(0, 2, 1, 3), kr.grad.float().numpy(), atol=2e-5)
    close(dv2.permute(0, 2, 1, 3), vr.grad.float().numpy(), atol=2e-5)
    dprobs2 = torch.empty_like(dprobs)
    assert ops.attn_capture_bwd(qc, kc, vc, probs, d_o.cuda(), dprobs2, scale, mode, need_dqkv=False) == (None, None, None)
    assert torch.equal(dprobs, dprobs2)


@pytest.mark.parametrize("B,H,N,D,shared,packed", [(4, 8, 77, 64, False, True), (3, 12, 50, 64, True, True), (2, 2, 14, 32, False, False),
                                                    (2, 3, 128, 48, False, True), (1, 4, 33, 8, False, False)])
def test_whole_head_backward_bf16_gradient_io(ops, B, H, N, D, shared, packed):
    """``attention_head.hip`` with bf16 gradient I/O (a bf16 ``d_o`` without ``mma_bf16``: the bf16 gradient stream of a bf16 body
    through a short tower): the SAME exact-fp32 arithmetic as the fp32-I/O launch on the same (bf16-representable) d_o, so dP is
    bit-identical and dq / dk / dv are the fp32 results rounded to bf16 once.  ``packed``: outputs are views of one [B, N, 3, H, D]
    tensor, as clip_model.backward_tape hands them in.  A long sequence without ``mma_bf16`` must refuse a bf16 d_o."""
    g = torch.Generator().manual_seed(N * 3 + D)
    Bf = 1 if shared else B
    qkv = torch.randn(Bf, N, 3, H, D, generator=g).cuda()
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    d_o16 = torch.randn(B, N, H, D, generator=g).to(torch.bfloat16).cuda()
    scale = D ** -0.5
    mask = torch.full((N, N), float("-inf")).triu_(1).cuda()
    probs = torch.empty(Bf, H, N, N, device="cuda")
    o = ops.attn_capture_fwd(q, k, v, probs, scale, 0, mask)
    res = {}
    for dt in (torch.float32, torch.bfloat16):
        d_o = d_o16.to(dt)
        dprobs = torch.full((B, H, N, N), float("nan"), device="cuda")
        if packed:
            dqkv = torch.full((B, N, 3, H, D), float("nan"), dtype=dt, device="cuda")
            out = (dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2])
        else:
            out = None
        got = ops.attn_capture_bwd(q, k, v, probs, d_o, dprobs, scale, 0, out=out, batch=B if shared else None, o=o)
        torch.cuda.synchronize()
        assert all(t.dtype == dt for t in got)
        res[dt] = (dprobs, [t.clone() for t in got])
    assert torch.equal(res[torch.float32][0], res[torch.bfloat16][0])                 # the captured gradient: same bits
    for a32, a16 in zip(res[torch.float32][1], res[torch.bfloat16][1]):
        assert torch.isfinite(a16.float()).all()
        assert torch.equal(a32.to(torch.bfloat16), a16)                                # one rounding of the same fp32 result
    if N == 77:
        long_q = torch.randn(1, 200, 3, 2, 64, device="cuda")
        p = torch.empty(1, 2, 200, 200, device="cuda")
        ops.attn_capture_fwd(long_q[:, :, 0], long_q[:, :, 1], long_q[:, :, 2], p, 0.125, 0, None)
        with pytest.raises(ops.MMXError):
            ops.attn_capture_bwd(long_q[:, :, 0], long_q[:, :, 1], long_q[:, :, 2], p,
                                 torch.randn(1, 200, 2, 64, device="cuda").to(torch.bfloat16), torch.empty_like(p), 0.125, 0)


@pytest.mark.parametrize("B,H,Nq,Nk,D,mode,masked", [(1, 8, 950, 950, 32, 0, False), (1, 8, 100, 950, 32, 0, False),
                                                     (1, 3, 37, 130, 20, 1, True), (2, 2, 16, 65, 32, 0, True),
                                                     (1, 1, 5, 300, 8, 0, False), (1, 12, 197, 197, 64, 1, False),
                                                     (1, 2, 70, 300, 48, 0, True), (2, 3, 130, 577, 64, 0, False)])
def test_attn_fwd_small_grid_split_kernel(attn_options, B, H, Nq, Nk, D, mode, masked):
    """``attn_fwd_split_kernel`` (16-row workgroups whose four waves split the keys: the shared forward of DETR's K-query pass)
    against the fp64 softmax and against the 64-row streaming kernel it replaces on small grids (option ``attn_fwd_split``)."""
    ops = attn_options
    ops.set_option("attn_head", 0)
    g = torch.Generator().manual_seed(Nq * 5 + Nk)
    q, k, v = (torch.randn(B, n, H, D, generator=g) for n in (Nq, Nk, Nk))
    scale = D ** -0.5 if mode == 0 else D ** 0.5
    mask = (torch.randn(B, 1, Nk, generator=g) > 1.0).float() * -10000.0 if masked else None
    p_ref, o_ref = torch_attention(*(t.permute(0, 2, 1, 3) for t in (q, k, v)), scale, mode,
                                   None if mask is None else mask[:, None])
    outs = []
    try:
        for split in (1, 0):
            ops.set_option("attn_fwd_split", split)
            probs = torch.full((B, H, Nq, Nk), float("nan"), device="cuda")
            o = ops.attn_capture_fwd(q.cuda(), k.cuda(), v.cuda(), probs, scale, mode, mask.cuda() if masked else None)
            close(probs, p_ref.float().numpy(), atol=2e-6)
            close(o.permute(0, 2, 1, 3), o_ref.float().numpy(), atol=1e-5)
            outs.append((probs, o))
    finally:
        ops.set_option("attn_fwd_split", 1)
    torch.testing.assert_close(outs[0][0], outs[1][0], rtol=0, atol=1e-6)       # the same softmax, other summation order
    torch.testing.assert_close(outs[0][1], outs[1][1], rtol=0, atol=2e-6)


@pytest.mark.parametrize("B,N,E", [(64, 77, 512), (3, 5, 8), (1, 1, 4), (128, 577, 1024)])
def test_rows_to_dense_and_rows_add(ops, B, N, E):
    """One row per sample scattered into a zeroed dense tensor / added to an existing one (the class / EOT token's gradient of a
    CLIP tower's top block): equal to torch's zeros + index_put and gather + add + index_put, bit for bit."""
    g = torch.Generator().manual_seed(B + N)
    vals = torch.randn(B, E, generator=g).cuda()
    rows = torch.randint(0, N, (B,), generator=g).cuda()
    ar = torch.arange(B, device="cuda")
    want = torch.zeros(B, N, E, device="cuda")
    want[ar, rows] = vals
    assert torch.equal(ops.rows_to_dense(vals, rows, N), want)
    dense = torch.randn(B, N, E, generator=g).cuda()
    want2 = dense.clone()
    want2[ar, rows] += vals
    assert torch.equal(ops.rows_add_(dense, rows, vals), want2)


@pytest.mark.parametrize("rows,E,bias", [(100, 256, True), (1, 256, True), (37, 48, False), (128, 64, True), (129, 256, True),
                                         (100, 512, True)])
def test_small_linear_routes_the_few_row_square_projections(ops, rows, E, bias):
    """``ops.small_linear`` (``mmx_linear_f32``: exact-fp32 MFMA, bias in the epilogue) against ``nn.Linear`` in float64; shapes
    outside its family (more than 128 rows, wider than 256) fall through to the module itself."""
    torch.manual_seed(rows + E)
    lin = torch.nn.Linear(E, E, bias=bias).cuda()
    x = torch.randn(1, rows, E, device="cuda")
    got = ops.small_linear(x, lin)
    want = torch.nn.functional.linear(x.double(), lin.weight.double(), lin.bias.double() if bias else None)
    assert got.shape == want.shape
    close(got, want.detach().float().cpu().numpy(), atol=5e-6)
    with torch.no_grad():
        lin.weight.mul_(2.0)                                             # in-place update: the cached transposed copy is refreshed
    want2 = torch.nn.functional.linear(x.double(), lin.weight.double(), lin.bias.double() if bias else None)
    close(ops.small_linear(x, lin), want2.detach().float().cpu().numpy(), atol=1e-5)
    xg = x.clone().requires_grad_(True)                                  # an input with a graph: the module itself, graph kept
    assert ops.small_linear(xg, lin).grad_fn is not None


@pytest.mark.parametrize("B,H,N,shared", [(1, 12, 197, False), (8, 12, 197, True), (3, 4, 33, False), (2, 2, 5, False), (1, 3, 1, False),
                                          (2, 12, 50, True), (2, 8, 260, False)])
def test_avg_heads_vecmat_fused_row_rule(ops, B, H, N, shared):
    """``mmx_avg_heads_vecmat``: ``base + x . mean_h clamp(G * A, 0)`` in two launches, against the oracle's head average followed
    by a float64 vector-matrix product; shared-forward slabs, ragged last column chunk / row chunk, 16-bit slabs, NaN policy."""
    g = torch.Generator().manual_seed(B * 100 + H + N)
    cam = torch.randn((1 if shared else B) * H, N, N, generator=g).softmax(-1)
    grad = torch.randn(B * H, N, N, generator=g)
    x, base = torch.rand(B, N, generator=g), torch.randn(B, N, generator=g)
    cam_full = cam.repeat(B, 1, 1) if shared else cam
    abar = onp.avg_heads_batched(cam_full.numpy(), grad.numpy(), B).astype(np.float64)
    want = base.double().numpy() + np.einsum("bi,bij->bj", x.double().numpy(), abar)
    got = ops.avg_heads_vecmat(x.cuda(), cam.cuda(), grad.cuda(), batch_size=B, shared_attn=shared, base=base.cuda())
    close(got, want.astype(np.float32), atol=1e-5)
    two = ops.chain_vecmat(x.cuda(), ops.avg_heads(cam.cuda(), grad.cuda(), batch_size=B, shared_attn=shared), base=base.cuda())
    close(got, two.cpu().numpy(), atol=1e-5)
    close(ops.avg_heads_vecmat(x.cuda(), cam.cuda(), grad.cuda(), batch_size=B, shared_attn=shared), (want - base.double().numpy()
                                                                                                    + x.double().numpy()).astype(np.float32), atol=1e-5)
    for dt in (torch.bfloat16, torch.float16):
        c16, g16 = cam.to(dt), grad.to(dt)
        c16_full = c16.repeat(B, 1, 1) if shared else c16
        ab = onp.avg_heads_batched(c16_full.float().numpy(), g16.float().numpy(), B).astype(np.float64)
        w16 = base.double().numpy() + np.einsum("bi,bij->bj", x.double().numpy(), ab)
        close(ops.avg_heads_vecmat(x.cuda(), c16.cuda(), g16.cuda(), batch_size=B, shared_attn=shared, base=base.cuda()),
              w16.astype(np.float32), atol=1e-5)
    if N > 2:
        gn = grad.clone()
        gn[0, 1, 2] = float("nan")                                  # NaN propagates like clamp() does: column 2 of sample 0
        out = ops.avg_heads_vecmat(x.cuda(), cam.cuda(), gn.cuda(), batch_size=B, shared_attn=shared, base=base.cuda()).cpu()
        assert torch.isnan(out[0, 2]) and torch.isnan(out).sum() == 1


@pytest.mark.parametrize("shape", [(4928, 2048), (7, 13), (1, 3), (3, 4)])
def test_quick_gelu_fused(ops, shape):
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(*shape, generator=g) * 3).cuda().requires_grad_(True)
    up = torch.randn(*shape, generator=g).cuda()
    y = ops.quick_gelu(x)
    y.backward(up)
    xr = x.detach().double().requires_grad_(True)
    yr = xr * torch.sigmoid(1.702 * xr)
    yr.backward(up.double())
    close(y, yr.detach().float().cpu().numpy(), atol=2e-6)
    close(x.grad, xr.grad.float().cpu().numpy(), atol=2e-6)


@pytest.mark.parametrize("B,N,E,shared", [(64, 50, 768, True), (3, 7, 12, True), (2, 5, 64, False)])
def test_layernorm_bwd_add(ops, B, N, E, shared):
    g = torch.Generator().manual_seed(B + N + E)
    xb = 1 if shared else B
    x = torch.randn(xb, N, E, generator=g).double()
    gamma, beta = torch.randn(E, generator=g).double(), torch.randn(E, generator=g).double()
    dy, d_res = torch.randn(B, N, E, generator=g), torch.randn(B, N, E, generator=g)
    xr = x.expand(B, N, E).clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr, (E,), gamma, beta, 1e-5).backward(dy.double())
    want = (xr.grad + d_res.double()).float().numpy()
    _, mean, rstd = torch.native_layer_norm(x.float().cuda(), (E,), gamma.float().cuda(), beta.float().cuda(), 1e-5)
    got = ops.layernorm_bwd_add(dy.cuda(), x.float().cuda(), mean, rstd, gamma.float().cuda(), d_res.cuda())
    close(got, want, atol=2e-5)
    got2 = ops.layernorm_bwd_add(dy.cuda(), x.float().cuda(), mean, rstd, gamma.float().cuda())
    close(got2, xr.grad.float().numpy(), atol=2e-5)


@pytest.mark.parametrize("B,N,E4", [(12, 9, 64), (3, 5, 16), (1, 7, 24), (37, 3, 8)])
def test_quick_gelu_bwd_bf16_gradient_stream(ops, B, N, E4):
    """bf16 dy / dx, fp32 pre-activation shared by the batch (sweep kernel for B >= 4, element kernel below): the fp32
    product rounded ONCE to bf16."""
    g = torch.Generator().manual_seed(B * 131 + N)
    m = (torch.randn(1, N, E4, generator=g) * 3).cuda()
    dy = torch.randn(B, N, E4, generator=g).cuda().to(torch.bfloat16)
    got = ops.quick_gelu_bwd(m, dy)
    assert got.dtype == torch.bfloat16 and got.shape == dy.shape
    md = m.double()
    sg = torch.sigmoid(1.702 * md)
    want = dy.double() * (sg + 1.702 * md * sg * (1 - sg))
    err = (got.double() - want).abs()
    assert float((err / want.abs().clamp_min(1e-30)).max()) <= 2 ** -8 + 1e-6       # half a bf16 ulp + the fp32 rounding of the factor
    assert torch.equal(ops.quick_gelu_bwd(m.expand(B, N, E4).contiguous(), dy), got)   # un-shared form: the same bits


@pytest.mark.parametrize("B,N,E,shared,res", [(8, 11, 1024, True, True), (3, 5, 768, False, True), (2, 4, 512, True, False),
                                              (2, 3, 1280, False, True), (3, 7, 96, True, True)])
def test_layernorm_bwd_add_bf16_gradient_stream(ops, B, N, E, shared, res):
    """bf16 upstream gradient, fp32 + bf16 results in one pass (row-resident kernel for E = 256 k, generic otherwise)."""
    g = torch.Generator().manual_seed(B + N + E)
    xb = 1 if shared else B
    x = torch.randn(xb, N, E, generator=g).double()
    gamma, beta = torch.randn(E, generator=g).double(), torch.randn(E, generator=g).double()
    dy = torch.randn(B, N, E, generator=g).to(torch.bfloat16)
    d_res = torch.randn(B, N, E, generator=g) if res else None
    xr = x.expand(B, N, E).clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr, (E,), gamma, beta, 1e-5).backward(dy.double())
    want = xr.grad + (d_res.double() if res else 0)
    _, mean, rstd = torch.native_layer_norm(x.float().cuda(), (E,), gamma.float().cuda(), beta.float().cuda(), 1e-5)
    dx, dx_h = ops.layernorm_bwd_add_bf16(dy.cuda(), x.float().cuda(), mean, rstd, gamma.float().cuda(),
                                          d_res.cuda() if res else None)
    close(dx, want.float().numpy(), atol=3e-5)
    assert torch.equal(dx_h, dx.to(torch.bfloat16))
    _, only_h = ops.layernorm_bwd_add_bf16(dy.cuda(), x.float().cuda(), mean, rstd, gamma.float().cuda(),
                                           d_res.cuda() if res else None, want_f32=False)
    assert torch.equal(only_h, dx_h)


def test_unsupported_shapes_fail_loudly(ops):
    """The C-ABI refuses what it does not implement (negative return code -> MMXError with the library's message);
    nothing falls back to another implementation."""
    from transformer_mm_explainability_amd import postprocess
    from transformer_mm_explainability_amd._lib import MMXError
    q = torch.randn(1, 5, 2, 80, device="cuda")                      # head_dim 80 > 64
    probs = torch.empty(1, 2, 5, 5, device="cuda")
    with pytest.raises(MMXError, match="head_dim"):
        ops.attn_capture_fwd(q, q, q, probs, 80 ** -0.5, 0, None)
    with pytest.raises(MMXError, match="64"):
        postprocess.image_heatmaps(torch.rand(1, 65 * 65, device="cuda"), 224)   # patch grid > 64
    big = [(torch.rand(1, 2, 60, 60, device="cuda").softmax(-1), torch.rand(1, 2, 60, 60, device="cuda"))]
    cross = [(torch.rand(1, 2, 60, 36, device="cuda").softmax(-1), torch.rand(1, 2, 60, 36, device="cuda"))]
    with pytest.raises(MMXError):
        ops.lxmert_schedule(big, big[:0], cross, [], big, [])          # T = 60 > 48: the one-launch schedule is LDS-sized
    with pytest.raises(MMXError, match="inner dims"):
        ops.matmul(torch.rand(4, 5, device="cuda"), torch.rand(4, 4, device="cuda"))


@pytest.mark.parametrize("dtype,rel", [(torch.bfloat16, 2 ** -8), (torch.float16, 2 ** -11)])
@pytest.mark.parametrize("B,H,Nq,Nk,D,masked", [(2, 4, 130, 130, 64, True), (1, 3, 70, 200, 32, False)])
def test_attn_capture_half_precision_slabs(attn_options, dtype, rel, B, H, Nq, Nk, D, masked):
    """fp16 / bf16 capture slabs (streaming kernels): P and dP equal the fp32 results rounded to the slab type; O comes
    from the unrounded P; dq/dk/dv are the exact gradients for the ROUNDED P (what the backward reads)."""
    g = torch.Generator().manual_seed(Nq + Nk)
    q, k, v, d_o = (torch.randn(B, n, H, D, generator=g).cuda() for n in (Nq, Nk, Nk, Nq))
    mask = torch.full((Nq, Nk), float("-inf")).triu_(1).cuda() if masked else None
    scale = D ** -0.5
    p32 = torch.empty(B, H, Nq, Nk, device="cuda")
    dp32 = torch.empty_like(p32)
    ops = attn_options                                                   # fp32 reference on the same (streaming) kernels
    ops.set_option("attn_fwd_split", 0)                                  # (half-precision slabs always take the 64-row forward)
    o32 = ops.attn_capture_fwd(q, k, v, p32, scale, 0, mask)
    ops.attn_capture_bwd(q, k, v, p32, d_o, dp32, scale, 0)
    ops.set_option("attn_fwd_split", 1)
    p16 = torch.empty(B, H, Nq, Nk, device="cuda", dtype=dtype)
    dp16 = torch.empty_like(p16)
    o16 = ops.attn_capture_fwd(q, k, v, p16, scale, 0, mask)
    dq, dk, dv = ops.attn_capture_bwd(q, k, v, p16, d_o, dp16, scale, 0)
    assert torch.equal(o16, o32) or (o16 - o32).abs().max() < 1e-6
    assert torch.equal(p16, p32.to(dtype))                               # same values, rounded to nearest even once
    torch.testing.assert_close(dp16.float(), dp32, rtol=2 * rel, atol=1e-6)
    # reference gradients for the rounded P: dS = P16 * (dP - rowsum(P16 * dP)), dQ = dS K scale, dK = dS^T Q scale, dV = P16^T dO
    P = p16.double()
    dP = torch.einsum("bqhd,bkhd->bhqk", d_o.double(), v.double())
    dS = P * (dP - (P * dP).sum(-1, keepdim=True))
    torch.testing.assert_close(dq.double(), torch.einsum("bhqk,bkhd->bqhd", dS, k.double()) * scale, rtol=1e-3, atol=5 * rel)
    torch.testing.assert_close(dv.double(), torch.einsum("bhqk,bqhd->bkhd", P, d_o.double()), rtol=1e-3, atol=1e-4)
    # the rule kernels read the half-precision slabs directly
    abar = ops.avg_heads(p16.view(B * H, Nq, Nk), dp16.view(B * H, Nq, Nk), batch_size=B)
    want = (p16.float() * dp16.float()).clamp(min=0).view(B, H, Nq, Nk).mean(1)
    torch.testing.assert_close(abar, want, rtol=1e-5, atol=1e-7)


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("N,B,shared", [(197, 3, False), (197, 4, True), (77, 2, False), (300, 1, False)])
def test_relevancy_chain_row_equals_the_matrix_chain(N, B, shared):
    """``ops.relevancy_chain_row`` (one row of R carried top-down as a vector: head average + mat-vec per layer) vs the
    rows of ``relevancy_self_chain`` (itself held against the oracle above), per-sample row indices, shared probabilities."""
    from transformer_mm_explainability_amd import ops
    g = torch.Generator().manual_seed(N + B)
    L, H = 4, 3
    attn = [torch.rand((1 if shared else B) * H, N, N, generator=g).softmax(-1).cuda() for _ in range(L)]
    grad = [(torch.randn(B * H, N, N, generator=g) * 0.05).cuda() for _ in range(L)]
    rows = torch.randint(0, N, (B,), generator=g).cuda()
    R = ops.relevancy_self_chain(attn, grad, B, shared_attn=shared)
    want = R[torch.arange(B, device="cuda"), rows]
    got = ops.relevancy_chain_row(attn, grad, B, rows, shared_attn=shared)
    assert got.shape == want.shape == (B, N)
    assert float((got - want).abs().max()) <= 2e-6 * float(want.abs().max())
    got0 = ops.relevancy_chain_row(attn, grad, B, 0, shared_attn=shared)
    assert float((got0 - R[:, 0]).abs().max()) <= 2e-6 * float(R[:, 0].abs().max())


@pytest.mark.parametrize("K,H,Q,Ni,L,shared", [(3, 4, 7, 15, 3, True), (2, 2, 33, 70, 2, False), (5, 8, 100, 950, 6, True)])
def test_detr_decoder_rows_kernels_vs_per_layer_ops(ops, K, H, Q, Ni, L, shared):
    """K2-DETR (``mmx_detr_decoder_rows``: three launches for all decoder layers) vs the per-layer formulation it replaces,
    written with torch ops in fp64 on the same slabs: rule 5 maps, R_qq^(l), eq. 8-9, w_l = u_l N(R_qq^(l))^T, z_l = w_l C_l,
    u_(l-1) = u_l (I + B_l), the NaN policy (a layer whose cross map holds a NaN contributes nothing to THAT sample) and
    the diag(R_qq - I) word.  DETR/modules/ExplanationGenerator.py:19-53, 120-140."""
    g = torch.Generator().manual_seed(K * 100 + Q)
    sm = lambda *s: torch.softmax(torch.randn(*s, generator=g), -1).cuda()          # noqa: E731
    gr = lambda *s: (torch.randn(*s, generator=g) * 0.3).cuda()                      # noqa: E731
    Ka = 1 if shared else K
    self_pairs = [(sm(Ka * H, Q, Q), gr(K * H, Q, Q)) for _ in range(L)]
    cross_pairs = [(sm(Ka * H, Q, Ni), gr(K * H, Q, Ni)) for _ in range(L)]
    poisoned = (K - 1, L // 2)
    cross_pairs[poisoned[1]][1].view(K, H, Q, Ni)[poisoned[0], 0, Q // 2, Ni // 3] = float("nan")
    targets = torch.randint(0, Q, (K,), generator=g).cuda()
    s, dmin = ops.detr_decoder_rows(self_pairs, cross_pairs, targets, shared_attn=shared)

    def cam(pair, nk):
        a, gg = pair
        a = a.double().view(Ka, H, Q, nk).expand(K, H, Q, nk)
        return (gg.double().view(K, H, Q, nk) * a).clamp(min=0).mean(1)             # [K, Q, nk]
    eye = torch.eye(Q, dtype=torch.float64, device="cuda")
    R = eye.repeat(K, 1, 1)
    Bs, hats, dm = [], [], []
    for l in range(L):
        B = cam(self_pairs[l], Q)
        R = R + B @ R
        Bs.append(B)
        hats.append((R - eye) / (R - eye).sum(-1, keepdim=True) + eye)
        dm.append((torch.diagonal(R, dim1=1, dim2=2) - 1).min())
    u = torch.zeros(K, 1, Q, dtype=torch.float64, device="cuda")
    u.scatter_(2, targets.reshape(K, 1, 1), 1.0)
    want = torch.zeros(K, 1, Ni, dtype=torch.float64, device="cuda")
    for l in range(L - 1, -1, -1):
        C = cam(cross_pairs[l], Ni)
        z = (u @ hats[l].transpose(1, 2)) @ torch.nan_to_num(C)
        clean = ~(torch.isnan(hats[l]).flatten(1).any(1) | torch.isnan(C).flatten(1).any(1))
        want = want + torch.where(clean.reshape(K, 1, 1), z, torch.zeros_like(z))
        u = u + u @ Bs[l]
    want = want.reshape(K, Ni)
    assert torch.isfinite(s).all()
    scale = float(want.abs().max())
    assert float((s.double() - want).abs().max()) <= 2e-6 * scale, float((s.double() - want).abs().max()) / scale
    assert abs(float(dmin) - float(torch.stack(dm).min())) <= 1e-6
    # the poisoned layer really was dropped for that sample only: recompute without the drop and compare
    assert float((s[poisoned[0]].double() - want[poisoned[0]]).abs().max()) <= 2e-6 * scale


def test_detr_decoder_rows_diag_word_carries_nan(ops):
    """a NaN on the diagonal of R_qq - I must reach the ``diag_min`` word (the reference's ``assert diag.min() >= 0``
    fails on NaN; ``fminf`` would drop it) so that ``MaskGenerator.check_diag`` / the deferred assert fire."""
    K, H, Q, Ni, L = 3, 2, 10, 40, 3
    g = torch.Generator().manual_seed(5)
    sm = lambda *s: torch.softmax(torch.randn(*s, generator=g), -1).cuda()          # noqa: E731
    gr = lambda *s: (torch.randn(*s, generator=g) * 0.3).cuda()                      # noqa: E731
    self_pairs = [(sm(K * H, Q, Q), gr(K * H, Q, Q)) for _ in range(L)]
    cross_pairs = [(sm(K * H, Q, Ni), gr(K * H, Q, Ni)) for _ in range(L)]
    targets = torch.randint(0, Q, (K,), generator=g).cuda()
    _, clean = ops.detr_decoder_rows(self_pairs, cross_pairs, targets, shared_attn=False)
    assert float(clean) == float(clean)                                              # finite on clean slabs
    self_pairs[1][1].view(K, H, Q, Q)[1, 0, 4, 4] = float("nan")                      # poisons diag(R_qq) of sample 1
    s, dmin = ops.detr_decoder_rows(self_pairs, cross_pairs, targets, shared_attn=False)
    assert torch.isnan(dmin).all(), "the NaN was dropped by the diag reduction"
    assert not (float(dmin) >= 0)                                                    # what the deferred assert evaluates
    # rules 6 / 7 have no scrub (ExplanationGenerator.py:27-30): a NaN self-attention map poisons that sample's rows, only that sample's
    assert torch.isnan(s[1]).any() and torch.isfinite(s[0]).all() and torch.isfinite(s[2]).all()


def test_graph_capture_holds_the_garbage_collector_off(ops):
    """Every ``Graphed*`` wrapper captures through ``ops.graph_capture``: Python's cyclic collector must not run inside a
    capture (it may destroy an earlier wrapper's CUDAGraph, whose pool release is illegal while a stream is capturing and
    aborts the process -- seen once in this suite), and its state is restored afterwards, also on an exception."""
    import gc

    class Cyclic:                                  # cyclic garbage holding a captured graph, like a dropped Graphed* wrapper
        def __init__(self):
            self.graph = torch.cuda.CUDAGraph()
            self.x = torch.zeros(1024, device="cuda")
            with ops.graph_capture(self.graph):
                self.y = self.x * 2 + 1
            self.me = self

    for _ in range(3):
        Cyclic()                                   # unreachable at once, only the cyclic collector frees it
    g = torch.cuda.CUDAGraph()
    x = torch.zeros(4, device="cuda")
    assert gc.isenabled()
    with ops.graph_capture(g):
        assert not gc.isenabled()
        junk = [[i] for i in range(5000)]          # enough allocations to trigger generation-0 collections if they were on
        y = x + len(junk)
    assert gc.isenabled()
    g.replay()
    torch.cuda.synchronize()
    assert float(y[0]) == 5000.0
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")            # (torch warns about the empty capture it ends on the way out)
        with pytest.raises(RuntimeError):
            with ops.graph_capture(torch.cuda.CUDAGraph()):
                raise RuntimeError("inside")
    assert gc.isenabled()
    gc.collect()


@pytest.mark.parametrize("B,H,T,I,nl,nv,nx,ragged", [
    (32, 12, 14, 36, 9, 5, 5, False),      # cfg 4
    (5, 4, 20, 36, 3, 2, 3, True),         # padded batch, per-sample question lengths
    (3, 2, 48, 48, 2, 2, 2, True),         # the LDS-resident limit
    (2, 3, 7, 12, 1, 1, 1, False),         # a single cross layer (no image side at all)
    (1, 12, 9, 11, 3, 2, 3, False),        # one item per call, like the reference's evaluator
    (70, 2, 6, 5, 2, 1, 2, True),          # more samples than one wave of workgroups per sample would fill
])
@pytest.mark.parametrize("flags", [{}, {"apply_normalization": False}, {"apply_self_in_rule_10": False}])
def test_lxmert_schedule_vs_oracle(ops, B, H, T, I, nl, nv, nx, ragged, flags):
    """``mmx_lxmert_schedule`` (chip-wide rule 5 + last-arriver schedule on the MFMA) against the oracle's restatement of the
    reference generator's schedule (``oracle/relevancy_np.lxmert_generate_ours_chain``, pinned on the reference's own outputs in
    tests/test_oracle_golden.py) sample by sample: per-sample question lengths (the rules of sample b run on its leading
    ``text_len[b]`` tokens, everything beyond is zero), both flags, all four matrices and the handle_residual word."""
    g = torch.Generator().manual_seed(B * 1000 + T * 10 + I)
    sm = lambda *s: torch.softmax(torch.randn(*s, generator=g), -1)                 # noqa: E731
    gr = lambda *s: torch.randn(*s, generator=g) * 0.2                               # noqa: E731
    pair = lambda nq, nk: (sm(B, H, nq, nk), gr(B, H, nq, nk))                       # noqa: E731
    groups = ([pair(T, T) for _ in range(nl)], [pair(I, I) for _ in range(nv)], [pair(T, I) for _ in range(nx)],
              [pair(I, T) for _ in range(nx - 1)], [pair(T, T) for _ in range(nx)], [pair(I, I) for _ in range(nx - 1)])
    text_len = torch.randint(1, T + 1, (B,), generator=g) if ragged else None
    if ragged:                # what the attention mask does to a padded batch: no probability on the padded keys / rows
        for grp, (tq, tk) in zip(groups, ((1, 1), (0, 0), (1, 0), (0, 1), (1, 1), (0, 0))):
            for a, _ in grp:
                for b in range(B):
                    t = int(text_len[b])
                    if tk:
                        a[b, :, :, t:] = 0
                        a[b] /= a[b].sum(-1, keepdim=True).clamp_min(1e-30)
                    if tq:
                        a[b, :, t:, :] = 0
    dev_groups = [[(a.cuda(), g_.cuda()) for a, g_ in grp] for grp in groups]
    got = ops.lxmert_schedule(*dev_groups, check_diag="defer", text_len=text_len.cuda() if ragged else None, **flags)
    torch.cuda.synchronize()
    kw = dict(normalize_self_attention=flags.get("apply_normalization", True), apply_self_in_rule_10=flags.get("apply_self_in_rule_10", True))
    for b in range(B):
        t = int(text_len[b]) if ragged else T
        cut = lambda pr, q, k: (pr[0][b, :, :(t if q else I), :(t if k else I)].numpy(),      # noqa: E731
                                pr[1][b, :, :(t if q else I), :(t if k else I)].numpy())
        la, lg = zip(*[cut(p, 1, 1) for p in groups[0]]) if nl else ((), ())
        va, vg = zip(*[cut(p, 0, 0) for p in groups[1]]) if nv else ((), ())
        x_layers = []
        for i in range(nx):
            blk = {"lang_cross": cut(groups[2][i], 1, 0), "lang_self": cut(groups[4][i], 1, 1)}
            if i < nx - 1:
                blk["img_cross"], blk["img_self"] = cut(groups[3][i], 0, 1), cut(groups[5][i], 0, 0)
            x_layers.append(blk)
        if not nl or not nv:
            continue                           # (the oracle reads the sizes off the first language / vision layer)
        want = onp.lxmert_generate_ours_chain(list(la), list(lg), list(va), list(vg), x_layers, return_all=True, **kw)
        for name, w, full, (q, k) in zip(("R_tt", "R_ti", "R_ii", "R_it"), want, got[:4], ((1, 1), (1, 0), (0, 0), (0, 1))):
            block = full[b, :(t if q else I), :(t if k else I)].cpu().numpy()
            scale = max(float(np.abs(w).max()), 1.0)
            assert np.array_equal(np.isnan(block), np.isnan(w)), name
            err = float(np.nanmax(np.abs(block - w))) if w.size else 0.0
            assert err <= 1e-5 * scale, (name, b, err, scale)
            rest = full[b].clone()
            rest[:(t if q else I), :(t if k else I)] = 0
            assert float(rest.abs().max()) == 0.0, (name, b, "entries beyond the sample's question length must be zero")
    if got[4] is not None:
        assert torch.isfinite(got[4]).all()


def test_lxmert_schedule_propagates_nan(ops):
    """LXMERT's rule 10 has no NaN scrub (lxmert/lxmert/src/ExplanationGenerator.py:32-42): a NaN in a slab must poison the
    sample it belongs to and nothing else."""
    B, H, T, I = 2, 3, 6, 8
    g = torch.Generator().manual_seed(11)
    sm = lambda *s: torch.softmax(torch.randn(*s, generator=g), -1).cuda()          # noqa: E731
    gr = lambda *s: (torch.randn(*s, generator=g) * 0.2).cuda()                      # noqa: E731
    pair = lambda nq, nk: (sm(B, H, nq, nk), gr(B, H, nq, nk))                       # noqa: E731
    groups = ([pair(T, T) for _ in range(2)], [pair(I, I) for _ in range(2)], [pair(T, I) for _ in range(2)],
              [pair(I, T) for _ in range(1)], [pair(T, T) for _ in range(2)], [pair(I, I) for _ in range(1)])
    groups[2][0][1][1, 0, 2, 3] = float("nan")            # sample 1, first language cross block
    out = ops.lxmert_schedule(*groups, check_diag="defer")
    for m in out[:4]:
        assert not torch.isnan(m[0]).any()                 # sample 0 is untouched
    assert torch.isnan(out[1][1]).any()                    # ... and sample 1's R_ti is poisoned


def test_bf16_outputs_of_layernorm_and_gelu_forward(ops):
    """Round 4: in a bf16 body the LayerNorm output and the MLP activation only feed a half-precision GEMM, so the kernels write
    them as bf16 directly (``add_layernorm(h_dtype=bfloat16)``, ``quick_gelu_fwd(out_dtype=bfloat16)``): must equal the fp32
    kernel's result rounded to nearest even, bit for bit; sum / statistics unchanged."""
    g = torch.Generator().manual_seed(3)
    for rows, E in ((5, 768), (77, 1024), (3, 3072)):
        x = torch.randn(rows, E, generator=g).cuda()
        y = torch.randn(rows, E, generator=g).cuda()
        gamma, beta = torch.randn(E, generator=g).cuda(), torch.randn(E, generator=g).cuda()
        s32, h32, m32, r32 = ops.add_layernorm(x, y, gamma, beta, 1e-5)
        s16, h16, m16, r16 = ops.add_layernorm(x, y, gamma, beta, 1e-5, h_dtype=torch.bfloat16)
        assert h16.dtype == torch.bfloat16 and torch.equal(h16, h32.to(torch.bfloat16))
        assert torch.equal(s16, s32) and torch.equal(m16, m32) and torch.equal(r16, r32)
        a32, a16 = ops.quick_gelu_fwd(x), ops.quick_gelu_fwd(x, torch.bfloat16)
        assert a16.dtype == torch.bfloat16 and torch.equal(a16, a32.to(torch.bfloat16))


@pytest.mark.parametrize("L,B,H,N,dt,shared,init", [
    (3, 2, 3, 197, torch.float32, False, False), (2, 3, 4, 577, torch.bfloat16, True, False), (2, 2, 2, 129, torch.float16, False, True),
    (3, 1, 5, 255, torch.float32, True, True), (2, 2, 2, 636, torch.float32, False, False), (2, 1, 3, 950, torch.float32, False, False),
    (1, 2, 2, 300, torch.bfloat16, False, False), (2, 1, 2, 1050, torch.bfloat16, False, True), (2, 2, 3, 639, torch.float32, False, False),
    (2, 90, 2, 197, torch.bfloat16, True, False), (2, 75, 1, 130, torch.float32, False, True)])    # > 256 blocks: several rounds of the persistent grid
def test_long_chain_one_launch_per_layer(chain_options, L, B, H, N, dt, shared, init):
    """``relevancy_chain_rows.hip`` (round 6: head reduction of a 16-row block into LDS + that block row of ``A_bar . R`` in one launch,
    ``self_chain_rows`` = 1; an opt-in: measured slower than the default two launches) == the numpy oracle's chain on the same (rounded) slabs and == the two-launch form
    (``avg_heads_kernel`` + tiled product) to summation order: fp32 / fp16 / bf16 slabs, shared probabilities, a given ``R_init``, odd
    sizes on both sides of every instantiation's limits (636 / 639: the second falls back to two launches by design)."""
    ops = chain_options
    g = torch.Generator().manual_seed(N + L)
    attn = [torch.rand((1 if shared else B) * H, N, N, generator=g).softmax(-1).to(dt) for _ in range(L)]
    grad = [(torch.randn(B * H, N, N, generator=g) * 0.05).to(dt) for _ in range(L)]
    r0 = (torch.eye(N) + torch.rand(B, N, N, generator=g) * 0.01) if init else None
    run = lambda: ops.relevancy_self_chain([a.cuda() for a in attn], [x.cuda() for x in grad], B,   # noqa: E731
                                           R_init=None if r0 is None else r0.cuda(), shared_attn=shared).cpu()
    ops.set_option("self_chain_rows", 1)
    got = run()
    ops.set_option("self_chain_rows", 0)
    two = run()
    a32 = [a.float().numpy().reshape(-1, H, N, N) for a in attn]
    g32 = [x.float().numpy().reshape(B, H, N, N) for x in grad]
    for b in range(B):
        R = np.eye(N, dtype=np.float32) if r0 is None else r0[b].numpy()
        for a, x in zip(a32, g32):
            cam = onp.avg_heads(a[0 if shared else b], x[b])
            R = R + cam @ R
        close(got[b], R)
        close(two[b], R)
    assert float((got - two).abs().max()) <= 2e-6 * float(two.abs().max())


@pytest.mark.parametrize("L,B,H,N,groups", [(12, 64, 8, 77, 0), (12, 160, 8, 77, 0), (5, 3, 3, 77, 2), (4, 2, 5, 128, 1), (3, 70, 2, 33, 0),
                                            (6, 9, 4, 50, 3), (2, 1, 1, 17, 1)])
def test_causal_flag_skips_the_upper_triangle_with_the_same_bits(chain_options, L, B, H, N, groups):
    """``MMX_CHAIN_CAUSAL`` (round 6): on the slabs of a causally masked tower (probabilities exactly 0 above the diagonal, gradients
    ANYTHING finite there) the chain kernels that stream fp32 slabs do not request the 4-element chunks that lie entirely above the
    diagonal -- and return the SAME BITS as the full read, for the layer-group kernel, the column kernel and every group count; the
    oracle on the same slabs agrees to 1e-5.  Non-causal slabs with the flag set would be wrong by design (the caller vouches)."""
    ops = chain_options
    attn, grad = make_layers(N * 3 + L, L, B, H, N, causal=True)
    attn, grad = [a.cuda() for a in attn], [g.cuda() for g in grad]
    for a in attn:
        assert float(a.triu(1).abs().max()) == 0.0
    ops.set_option("self_chain_groups", groups)
    full = ops.relevancy_self_chain(attn, grad, B).clone()
    skip = ops.relevancy_self_chain(attn, grad, B, causal=True).clone()
    assert torch.equal(full, skip)
    plan = ops.ChainPlan(attn, grad, B, causal=True)
    assert torch.equal(plan.launch(), full)
    # a caller-supplied start is not triangular: the tile shortcut of the causal form (lower-triangular R) must stay off
    R0 = torch.randn(B, N, N, device="cuda", generator=torch.Generator(device="cuda").manual_seed(N))
    assert torch.equal(ops.relevancy_self_chain(attn, grad, B, R_init=R0, causal=True), ops.relevancy_self_chain(attn, grad, B, R_init=R0))
    close(full, onp.self_chain([a.cpu().numpy() for a in attn], [g.cpu().numpy() for g in grad], B))
    # the flag is honoured, not ignored: probabilities planted in chunks that lie ENTIRELY above the diagonal (16-byte aligned in the
    # flat slab, inside one row) change the full read and do not reach the causal one
    if N >= 33:
        bad = [a.clone() for a in attn]
        hot = [g.clone() for g in grad]
        for row in (2, 5):
            c0 = next(c for c in range(row + 1, N) if (row * N + c) % 4 == 0)
            assert c0 + 7 < N
            for t in (bad[0], hot[0]):
                t.reshape(B, H, N, N)[0, :, row, c0:c0 + 8] = 0.25       # two whole chunks, every head, gradient positive there too
        dirty, clean = ops.relevancy_self_chain(bad, hot, B).clone(), ops.relevancy_self_chain(attn, hot, B).clone()
        assert not torch.equal(dirty, clean)
        got = ops.relevancy_self_chain(bad, hot, B, causal=True)
        # (the flag is a licence, not an order: a route that does not stream fp32 slabs through chain_stream.h reads everything)
        assert torch.equal(got, clean) or (N < 50 and torch.equal(got, dirty))


@pytest.mark.parametrize("B,H,Nq,Nk,D,mask_kind", [
    (4, 8, 77, 77, 64, "causal"), (2, 12, 50, 50, 64, "none"), (3, 4, 128, 128, 32, "causal"), (2, 3, 33, 33, 16, "causal"),
    (2, 12, 14, 36, 64, "padding"), (3, 2, 100, 100, 64, "padding"), (2, 4, 77, 77, 64, "band"), (1, 2, 17, 120, 48, "padding"),
    (2, 2, 77, 77, 64, "dead_rows"),
])
def test_whole_head_masked_tile_skip_same_bits(attn_options, B, H, Nq, Nk, D, mask_kind):
    """``attention_head.hip`` skips the products of 16-key tiles that are masked out for a whole 16-row strip (causal towers: 10 of 25
    tiles at 77 tokens; padded batches; any additive -inf mask): P, O, dP, dq, dk, dv must be BIT-identical to the kernels with the skip
    switched off (option ``attn_head_tile_skip``), the results of the forward must still match the torch reference, and dP stays dense
    (the reference exposes it as ``attn_grad``: dO.V^T above the diagonal too).  ``band``: live tiles that are neither a prefix nor a
    suffix for some strips; ``dead_rows``: whole query rows masked (NaN rows in the reference too) must come out the same either way."""
    ops = attn_options
    g = torch.Generator().manual_seed(Nq * 11 + Nk)
    q, k, v, d_o = (torch.randn(B, n, H, D, generator=g).cuda() for n in (Nq, Nk, Nk, Nq))
    ninf = float("-inf")
    mask = None
    if mask_kind == "causal":
        mask = torch.full((Nq, Nk), ninf).triu_(1)
    elif mask_kind == "padding":
        lens = torch.randint(1, Nk + 1, (B,), generator=g)
        mask = torch.zeros(B, 1, Nk)
        for b in range(B):
            mask[b, 0, int(lens[b]):] = ninf
    elif mask_kind == "band":
        i, j = torch.arange(Nq)[:, None], torch.arange(Nk)[None, :]
        mask = torch.where((j > i) | (j < i - 20), torch.tensor(ninf), torch.tensor(0.0))
    elif mask_kind == "dead_rows":
        mask = torch.full((Nq, Nk), ninf).triu_(1)
        mask[5] = ninf
        mask[40:44] = ninf
    mc = mask.cuda() if mask is not None else None
    scale = D ** -0.5
    outs = {}
    for skip in (0, 1):
        ops.set_option("attn_head_tile_skip", skip)
        probs, dprobs = torch.full((B, H, Nq, Nk), 7.0, device="cuda"), torch.full((B, H, Nq, Nk), 7.0, device="cuda")
        o = ops.attn_capture_fwd(q, k, v, probs, scale, 0, mc)
        dq, dk, dv = ops.attn_capture_bwd(q, k, v, probs, d_o, dprobs, scale, 0)
        outs[skip] = [t.clone() for t in (probs, o, dprobs, dq, dk, dv)]
    for name, a, b in zip(("P", "O", "dP", "dq", "dk", "dv"), outs[0], outs[1]):
        assert torch.equal(a.nan_to_num(nan=123.0), b.nan_to_num(nan=123.0)) and torch.equal(a.isnan(), b.isnan()), name
    if mask_kind != "dead_rows":
        mref = mask if mask is None or mask.dim() == 2 else mask[:, None]
        qr, kr, vr = (t.cpu().permute(0, 2, 1, 3) for t in (q, k, v))
        p_ref, o_ref = torch_attention(qr, kr, vr, scale, 0, mref)
        close(outs[1][0], p_ref.float().numpy(), atol=2e-6)
        close(outs[1][1].permute(0, 2, 1, 3), o_ref.float().numpy(), atol=1e-5)
    if mask_kind == "causal":       # dP is dense: the gradient of the probabilities above the diagonal is dO.V^T, not zero
        assert float(outs[1][2][:, :, 0, Nk - 1].abs().min()) > 0.0
