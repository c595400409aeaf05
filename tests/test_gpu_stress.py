"""-m gpu: randomised stress + replay bit-stability of the hand-synchronised kernels.

GPU AddressSanitizer and XNACK builds are refused on the pool this suite runs on (``gpurun`` turns XNACK-on code objects and the
XNACK runtime mode down), so what a sanitizer pass would look for is approached from the outside instead:
  * seeded random shapes (>= 50 per kernel family) through every dispatch route of the chain kernels and the tiled product --
    strict-order routes bit-identical to each other, re-associated (layer-group) routes within 1e-5, the product vs fp64;
  * the same launches under UNEVEN load (a copy stream hammering HBM on a side stream while the chain runs), the regime in which a
    missing release / acquire of an in-launch hand-off shows (cdna guide G16: "test every hand-off under uneven load");
  * 200 replays of the graphed headline step and of the bi-modal schedule kernel (tickets + write-through partial products):
    every replay bit-identical to the first;
  * guard bands: outputs live inside larger NaN-filled allocations and the bands must come back untouched (a store past the end),
    inputs are followed by NaN poison that must not reach the result (a load past the end that is used).
The C-ABI argument checks run under the host AddressSanitizer on the CPU (``make -C csrc -f Makefile.asan asan-host``, tests/test_abi.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from transformer_mm_explainability_amd import ops as _ops
    return _ops


@pytest.fixture
def chain_defaults(ops):
    yield
    for k, v in (("self_chain_algo", 0), ("self_chain_groups", 0), ("bmm_tiles", 1), ("debug_flags", 0)):
        ops.set_option(k, v)


def _set(ops, algo, groups):
    ops.set_option("self_chain_algo", algo)
    ops.set_option("self_chain_groups", groups)


class _Hammer:
    """A side stream that keeps ~all of HBM busy with copies while the block runs (uneven load for the in-launch hand-offs)."""

    def __init__(self, on):
        self.on = on
        if on:
            self.stream = torch.cuda.Stream()
            self.src = torch.empty(96 << 20, dtype=torch.uint8, device="cuda")
            self.dst = torch.empty_like(self.src)

    def kick(self, n=6):
        if self.on:
            with torch.cuda.stream(self.stream):
                for _ in range(n):
                    self.dst.copy_(self.src, non_blocking=True)


def _chain_case(ops, rng, loaded):
    L = int(rng.integers(1, 15))
    B = int(rng.integers(1, 71))
    H = int(rng.integers(1, 13))
    N = int(rng.integers(17, 129))
    shared, with_init = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(0, 2 ** 31)))
    attn = [torch.rand((1 if shared else B) * H, N, N, device="cuda", generator=g).softmax(-1) for _ in range(L)]
    grad = [torch.randn(B * H, N, N, device="cuda", generator=g) * 0.05 for _ in range(L)]
    r0 = (torch.eye(N, device="cuda") + torch.rand(B, N, N, device="cuda", generator=g) * 0.1) if with_init else None
    hammer = _Hammer(loaded)

    def run():
        hammer.kick()
        return ops.relevancy_self_chain(attn, grad, B, R_init=r0, shared_attn=shared).clone()

    tag = "L=%d B=%d H=%d N=%d shared=%d init=%d loaded=%d" % (L, B, H, N, shared, with_init, loaded)
    _set(ops, 1, 1)
    ref = run()                                                    # fused kernel, strict layer order
    _set(ops, 5, 0)
    assert torch.equal(run(), ref), "cols kernel != strict order: " + tag
    _set(ops, 0, 1)
    assert torch.equal(run(), ref), "auto algo, one group != strict order: " + tag
    _set(ops, 0, 0)
    err = float((run() - ref).abs().max())
    assert err <= 1e-5, "auto route: %.2e: %s" % (err, tag)
    for G in (2, 3, 4):
        if G > L:
            continue
        _set(ops, 1, G)
        fused = run()
        _set(ops, 0, G)
        for _ in range(2):                                         # twice: arrival order differs between launches, bits must not
            assert torch.equal(run(), fused), "groups kernel != fused kernel at G=%d: %s" % (G, tag)
        err = float((fused - ref).abs().max())
        assert err <= 1e-5, "G=%d vs strict order %.2e: %s" % (G, err, tag)
    _set(ops, 0, 0)
    return err


@pytest.mark.parametrize("seed,cases,loaded", [(0, 40, False), (1, 20, True)])
def test_chain_kernels_random_shapes(ops, chain_defaults, seed, cases, loaded):
    """60 seeded random (layers, batch, heads, tokens, shared / per-sample probabilities, R_init) through every chain route
    (``tools/stress_chain.py`` as a test): strict-order routes bit-identical, layer-group routes bit-identical to the fused kernel at
    the same group count (whatever the arrival order) and within 1e-5 of strict order; a third of them under a saturating copy load."""
    rng = np.random.default_rng(seed)
    worst = max(_chain_case(ops, rng, loaded) for _ in range(cases))
    from parity import note
    note("layer-group routes vs strict order, worst of %d random shapes%s" % (cases, " under load" if loaded else ""), worst, 1e-5, 1.0)


def test_bmm_tiles_random_shapes(ops, chain_defaults):
    """50 seeded random (batch, M, N, K) products on ``bmm_f32_tiles.hip`` vs fp64 and vs the general kernel, with / without the
    additive input and the NaN scrub (``tools/stress_bmm.py`` as a test)."""
    rng = np.random.default_rng(7)
    worst = 0.0
    for case in range(50):
        M, N = int(rng.integers(96, 1101)), int(rng.integers(96, 1101))
        K = int(rng.integers(32, 1101))
        B = max(1, min(int(rng.integers(1, 41)), (256 * 64 * 64 * 4) // (M * N) + 1))
        cin, nan = bool(rng.integers(0, 2)), bool(rng.integers(0, 4) == 0)
        g = torch.Generator(device="cuda").manual_seed(case)
        a = torch.rand(B, M, K, device="cuda", generator=g) / K
        b = torch.randn(B, K, N, device="cuda", generator=g)
        c = torch.randn(B, M, N, device="cuda", generator=g) if cin else None
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
        tag = "B=%d M=%d N=%d K=%d cin=%d nan=%d" % (B, M, N, K, cin, nan)
        assert e64 <= 2e-6 and eold <= 2e-6, "%s: vs fp64 %.1e, vs general kernel %.1e" % (tag, e64, eold)
        assert nan or bool(torch.isfinite(got).all()), tag
        worst = max(worst, e64)
    from parity import note
    note("bmm_f32_tiles vs fp64, worst of 50 random shapes (relative to max |C|)", worst, 2e-6, 1.0)


def test_headline_step_200_replays_bit_stable():
    """The graphed cfg-2 step (CLIP ViT-B/32, B = 64, all 12 + 12 layers: library GEMMs, whole-head attention kernels, both chain
    launches with their tickets and write-through partial products) replayed 200 times: every replay's maps equal the first, bit for
    bit -- a hand-off that is occasionally stale, or a combine whose order follows arrival, cannot pass this."""
    from transformer_mm_explainability_amd import clip_explainability as ce
    from transformer_mm_explainability_amd import clip_model
    torch.manual_seed(0)
    model = clip_model.random_init("ViT-B/32", seed=0).cuda()
    g = torch.Generator().manual_seed(5)
    image = torch.randn(1, 3, 224, 224, generator=g).cuda()
    texts = torch.zeros(64, 77, dtype=torch.long)
    for b in range(64):
        n = 3 + b % 8
        texts[b, 0] = 49406
        texts[b, 1:1 + n] = torch.randint(1, 49405, (n,), generator=g)
        texts[b, 1 + n] = 49407
    texts = texts.cuda()
    run = ce.GraphedInterpret(model, image, texts, start_layer=0, start_layer_text=0)
    hammer = _Hammer(True)
    first = [t.clone() for t in run(image, texts)]
    torch.cuda.synchronize()
    bad = 0
    for i in range(200):
        if i % 3 == 0:
            hammer.kick(2)                      # every third replay runs beside a saturating copy stream
        R_text, R_image = run(image, texts)
        bad += int(not (torch.equal(R_text, first[0]) and torch.equal(R_image, first[1])))
    torch.cuda.synchronize()
    assert bad == 0, "%d of 200 replays differ from the first" % bad
    assert torch.isfinite(first[0]).all() and torch.isfinite(first[1]).all()


def test_lxmert_schedule_200_launches_bit_stable(ops):
    """``lxmert_schedule_v2_kernel`` (phase 1 spread over the chip with write-through A-bar blocks, a ticket per sample, the last
    arriver runs the 38 rule applications): 200 launches on the same slabs, half of them beside a copy stream, all bit-identical."""
    B, H, T, I = 32, 12, 14, 36
    g = torch.Generator().manual_seed(3)
    sm = lambda *s: torch.softmax(torch.randn(*s, generator=g), -1).cuda()          # noqa: E731
    gr = lambda *s: (torch.randn(*s, generator=g) * 0.2).cuda()                      # noqa: E731
    pair = lambda nq, nk: (sm(B, H, nq, nk), gr(B, H, nq, nk))                       # noqa: E731
    groups = ([pair(T, T) for _ in range(9)], [pair(I, I) for _ in range(5)], [pair(T, I) for _ in range(5)],
              [pair(I, T) for _ in range(4)], [pair(T, T) for _ in range(5)], [pair(I, I) for _ in range(4)])
    text_len = torch.randint(4, T + 1, (B,), generator=g).cuda()
    hammer = _Hammer(True)
    first = [t.clone() for t in ops.lxmert_schedule(*groups, check_diag="defer", text_len=text_len)[:4]]
    bad = 0
    for i in range(200):
        if i % 2:
            hammer.kick(1)
        out = ops.lxmert_schedule(*groups, check_diag="defer", text_len=text_len)[:4]
        bad += int(not all(torch.equal(a, b) for a, b in zip(out, first)))
    torch.cuda.synchronize()
    assert bad == 0, "%d of 200 launches differ from the first" % bad


POISON = float("nan")


def _banded(shape, dtype=torch.float32, band=4096):
    """A tensor of ``shape`` carved out of the middle of a NaN-filled allocation -> (view, whole buffer, band elements)."""
    n = int(np.prod(shape))
    whole = torch.full((n + 2 * band,), POISON, dtype=dtype, device="cuda")
    return whole[band:band + n].view(*shape), whole, band


def _bands_intact(whole, band, n):
    return bool(torch.isnan(whole[:band]).all()) and bool(torch.isnan(whole[band + n:]).all())


@pytest.mark.parametrize("L,B,H,N,groups", [(12, 64, 8, 77, 0), (12, 64, 12, 50, 0), (5, 3, 5, 33, 2), (7, 9, 3, 127, 1), (3, 2, 4, 21, 3)])
def test_chain_reads_and_writes_stay_inside_their_buffers(ops, chain_defaults, L, B, H, N, groups):
    """Guard bands around every slab of a chain launch: each probability / gradient slab ends right in front of NaN poison and the
    result must not change (an over-read that is USED would turn entries into NaN -- the odd-N^2 tail chunk of the stream waves reads
    through a buffer resource that ends at the tensor end), and the bands around R_out must come back untouched."""
    g = torch.Generator(device="cuda").manual_seed(N)
    plain_a = [torch.rand(B * H, N, N, device="cuda", generator=g).softmax(-1) for _ in range(L)]
    plain_g = [torch.randn(B * H, N, N, device="cuda", generator=g) * 0.05 for _ in range(L)]
    ops.set_option("self_chain_groups", groups)
    want = ops.relevancy_self_chain(plain_a, plain_g, B).clone()
    banded_a, banded_g = [], []
    for a, gr in zip(plain_a, plain_g):
        va, _, _ = _banded(a.shape)
        vg, _, _ = _banded(gr.shape)
        va.copy_(a)
        vg.copy_(gr)
        banded_a.append(va)
        banded_g.append(vg)
    got = ops.relevancy_self_chain(banded_a, banded_g, B)
    assert torch.equal(got, want)
    assert not torch.isnan(got).any()


def test_avg_heads_and_product_outputs_leave_guard_bands_alone(ops):
    """Output guard bands for the two kernels of the N > 128 chain (head reduction, tiled exact-fp32 product) at odd sizes."""
    from transformer_mm_explainability_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(9)
    for (B, H, N) in ((2, 3, 197), (1, 5, 131), (3, 2, 577)):
        a = torch.rand(B * H, N, N, device="cuda", generator=g).softmax(-1)
        gr = torch.randn(B * H, N, N, device="cuda", generator=g) * 0.05
        want = ops.avg_heads(a, gr, batch_size=B)
        out, whole, band = _banded((B, N, N))
        rc = _lib.lib().mmx_avg_heads(a.data_ptr(), gr.data_ptr(), out.data_ptr(), B, H, N, N, _lib.MMX_F32,
                                      torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.equal(out, want) and _bands_intact(whole, band, B * N * N)
        x = torch.randn(B, N, N, device="cuda", generator=g)
        wantp = ops.matmul(want, x)
        outp, wholep, bandp = _banded((B, N, N))
        nn = N * N
        rc = _lib.lib().mmx_bmm_f32(want.data_ptr(), x.data_ptr(), None, outp.data_ptr(), B, N, N, N, 0, nn, nn, nn, 0,
                                    torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.equal(outp, wantp) and _bands_intact(wholep, bandp, B * nn)


def test_whole_head_attention_random_shapes_and_masks(ops):
    """40 seeded random (batch, heads, Nq, Nk <= 128, head_dim, mask) through the whole-head attention kernels (attention_head.hip): the
    masked-tile skip (a wave-uniform switch over per-tile-count instantiations, live-strip words in LDS) on vs off must be BIT-identical
    for P, O, dP, dq, dk, dv -- every tile count 1..8 and every strip count gets exercised -- and both must agree with the general tiled
    kernels (another code path: attention_kernels.hip) to rounding level."""
    rng = np.random.default_rng(11)
    ninf = float("-inf")
    worst = 0.0
    try:
        for case in range(40):
            B, H = int(rng.integers(1, 4)), int(rng.integers(1, 5))
            Nq, Nk = int(rng.integers(1, 200)), int(rng.integers(1, 129))
            D = int(rng.choice([8, 16, 20, 32, 48, 64]))
            kind = ["none", "causal", "padding", "random_rows"][case % 4]
            if kind == "causal":
                Nq = Nk = min(Nq, Nk)
            g = torch.Generator(device="cuda").manual_seed(1000 + case)
            q, k, v, d_o = (torch.randn(B, n, H, D, device="cuda", generator=g) for n in (Nq, Nk, Nk, Nq))
            mask = None
            if kind == "causal":
                mask = torch.full((Nq, Nk), ninf, device="cuda").triu_(1)
            elif kind == "padding":
                mask = torch.zeros(B, 1, Nk, device="cuda")
                for b in range(B):
                    mask[b, 0, int(rng.integers(1, Nk + 1)):] = ninf
            elif kind == "random_rows":      # per-row visible window [lo, hi): live tiles anywhere, never an empty row
                lo = torch.from_numpy(rng.integers(0, Nk, Nq)).cuda()
                hi = torch.minimum(lo + torch.from_numpy(rng.integers(1, Nk + 1, Nq)).cuda(), torch.tensor(Nk, device="cuda"))
                j = torch.arange(Nk, device="cuda")[None, :]
                mask = torch.where((j >= lo[:, None]) & (j < hi[:, None]), 0.0, ninf).float()
            scale = D ** -0.5
            outs = {}
            for mode in ("skip", "noskip", "tiled"):
                ops.set_option("attn_head", int(mode != "tiled"))
                ops.set_option("attn_stream", int(mode != "tiled"))
                ops.set_option("attn_head_tile_skip", int(mode == "skip"))
                probs, dprobs = torch.empty(B, H, Nq, Nk, device="cuda"), torch.empty(B, H, Nq, Nk, device="cuda")
                o = ops.attn_capture_fwd(q, k, v, probs, scale, 0, mask)
                dq, dk, dv = ops.attn_capture_bwd(q, k, v, probs, d_o, dprobs, scale, 0)
                outs[mode] = [t.clone() for t in (probs, o, dprobs, dq, dk, dv)]
            for name, a, b, c in zip(("P", "O", "dP", "dq", "dk", "dv"), outs["skip"], outs["noskip"], outs["tiled"]):
                assert torch.equal(a, b), (case, kind, (B, H, Nq, Nk, D), name)
                err = float((a - c).abs().max()) / max(float(c.abs().max()), 1e-6)
                worst = max(worst, err)
                assert err <= 2e-5, (case, kind, (B, H, Nq, Nk, D), name, err)
    finally:
        for key in ("attn_head", "attn_stream", "attn_head_tile_skip"):
            ops.set_option(key, 1)
    from parity import note
    note("whole-head attention (skip on) vs the general tiled kernels, worst of 40 random shapes (relative to max |ref|)", worst, 2e-5, 1.0)
