"""CPU suite: the LRP pass of the LXMERT / VisualBERT bodies (``bert_lrp.py`` + the ``relprop`` methods of ``lxmert_model`` /
``visualbert_model``) against the reference's REAL pass (``lxmert_lrp.py`` / ``BERT_ours.py`` over their LRP layer library;
fixtures ``lxmert_model_lrp.npz`` / ``visualbert_model_lrp.npz`` from ``tests/golden/make_golden.py``).

No GPU here: the capture op of the attention modules is replaced by a plain-torch stand-in (test infrastructure, below) and the
attention core of the pass runs on the referee ``oracle/lrp_torch.py``; what is pinned is the host logic -- tapes, rule order,
Clone / Add / Linear closed forms, per-sample sums.  The HIP kernels are pinned on the same fixtures in ``tests/test_gpu_lrp.py``.

Tolerances: the pass divides by layer outputs (``safe_divide``), which makes the reference's OWN fp32 pass uncertain at up to
~2e-3 of a cam's largest entry -- measured: the fixtures also hold the same reference pass run in float64 (``f64__`` entries).
The yardstick is that distance: a result must be as close to the reference's float64 values as the reference's float32 values
are, within a factor (``within_reference_noise``).  (On the CPU these closed forms reproduce the reference's fp32 values bit
for bit in this image; the bound does not rely on it.)"""

import numpy as np
import pytest
import torch

from oracle import lrp_torch as lrp_oracle
from transformer_mm_explainability_amd import attention_modules, bert_lrp


def within_reference_noise(got, g, key, factor=4.0, floor=1e-5, what=None):
    """``|got - ref64| <= factor * |ref32 - ref64| + floor * max|ref64|`` (max norms), refs = the reference's pass in fp32 / fp64."""
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    got, r32, r64 = (np.asarray(x, dtype=np.float64) for x in (got, g[key], g["f64__" + key]))
    assert got.shape == r64.shape, (what or key, got.shape, r64.shape)
    top = float(np.abs(r64).max())
    err, noise = float(np.abs(got - r64).max()), float(np.abs(r32 - r64).max())
    assert err <= factor * noise + floor * max(top, 1e-30), \
        "%s: |got - ref64| %.3e > %.1f x the reference's own fp32 noise %.3e (+ %.1e of max %.3e)" % (what or key, err, factor,
                                                                                                      noise, floor, top)
    return err, noise, top


def capture_stand_in(q, k, v, probs, grads, scale, mask=None, scale_mode=None, **_):
    """What the HIP capture op computes (lxmert_lrp.py:385-420), in torch ops, with the hook that keeps dL/dP."""
    s = torch.einsum("bqhd,bkhd->bhqk", q, k) / scale
    if mask is not None:
        s = s + mask.reshape(mask.shape[0], 1, 1, -1)
    p = torch.softmax(s, dim=-1)
    probs.copy_(p.detach())
    if p.requires_grad:
        p.register_hook(lambda g: grads.copy_(g))
    return torch.einsum("bhqk,bkhd->bqhd", p, v)


@pytest.fixture
def cpu_body(monkeypatch):
    monkeypatch.setattr(attention_modules, "attention_capture", capture_stand_in)
    monkeypatch.setattr(attention_modules._SlabOwner, "_slabs",
                        lambda self, B, H, Nq, Nk, device: (torch.empty(B, H, Nq, Nk), torch.zeros(B, H, Nq, Nk)))


def t(x):
    return torch.from_numpy(np.asarray(x))


def lxmert_from_golden(g):
    from transformer_mm_explainability_amd import lxmert_model as lm
    hidden, heads, inter, ll, xl, rl, feat, vocab, labels, max_pos, T, I = (int(x) for x in g["dims"])
    cfg = lm.LxmertConfig(hidden_size=hidden, num_attention_heads=heads, intermediate_size=inter, l_layers=ll, x_layers=xl,
                          r_layers=rl, visual_feat_dim=feat, vocab_size=vocab, num_qa_labels=labels,
                          max_position_embeddings=max_pos)
    model = lm.LxmertForQuestionAnswering(cfg)
    model.load_state_dict({k[3:]: t(v) for k, v in g.items() if k.startswith("w__")}, strict=True)
    return model.eval(), {k[4:]: t(v) for k, v in g.items() if k.startswith("in__")}


def lxmert_cams(model):
    enc = model.lxmert.encoder
    cams = {}
    for i, b in enumerate(enc.layer):
        cams["l%d" % i] = b.attention.self
    for i, b in enumerate(enc.r_layers):
        cams["r%d" % i] = b.attention.self
    for i, b in enumerate(enc.x_layers):
        cams["x%d_lang_self" % i] = b.lang_self_att.self
        cams["x%d_visn_self" % i] = b.visn_self_att.self
        cams["x%d_cross" % i] = b.visual_attention.att
        cams["x%d_cross_copy" % i] = b.visual_attention_copy.att
    return cams


def test_lxmert_relprop_matches_reference_pass(golden, cpu_body):
    gm, g = golden("lxmert_model"), golden("lxmert_model_lrp")
    model, inputs = lxmert_from_golden(gm)
    out = model(**inputs).question_answering_score
    np.testing.assert_allclose(out.detach().numpy(), gm["score"], atol=1e-5)
    one_hot = torch.zeros_like(out)
    one_hot[0, int(g["index"])] = 1
    torch.sum(one_hot * out).backward()
    cam_lang, cam_vis = model.relprop(one_hot.clone(), alpha=1, core=lrp_oracle.core)
    for name, module in lxmert_cams(model).items():
        within_reference_noise(module.get_attn_cam(), g, "cam__" + name)
    within_reference_noise(cam_lang, g, "cam_lang")
    within_reference_noise(cam_vis, g, "cam_vis")
    # the top x-layer's image stream carries no relevance (the answer reads the language stream): zeros, as in the reference
    assert float(np.abs(g["cam__x2_cross_copy"]).max()) == 0.0
    assert float(model.lxmert.encoder.x_layers[-1].visual_attention_copy.att.get_attn_cam().abs().max()) == 0.0


def test_lxmert_relprop_is_per_sample(golden, cpu_body):
    """A batch gives what the reference's one-sample pass gives item by item (its whole-tensor sums are per sample here)."""
    gm, g = golden("lxmert_model"), golden("lxmert_model_lrp")
    model, inputs = lxmert_from_golden(gm)
    gen = torch.Generator().manual_seed(5)
    two = {k: torch.cat((v, v)) for k, v in inputs.items()}
    two["visual_feats"] = torch.cat((inputs["visual_feats"], torch.randn(inputs["visual_feats"].shape, generator=gen)))
    out = model(**two).question_answering_score
    one_hot = torch.zeros_like(out)
    one_hot[0, int(g["index"])] = 1
    one_hot[1, 3] = 1
    torch.sum(one_hot * out).backward()
    cam_lang, _ = model.relprop(one_hot.clone(), alpha=1, core=lrp_oracle.core)
    within_reference_noise(cam_lang[:1], g, "cam_lang", what="cam_lang of item 0 in a batch of 2")
    within_reference_noise(model.lxmert.encoder.layer[0].attention.self.get_attn_cam()[:1], g, "cam__l0", what="l0 of item 0")


def visualbert_from_golden(g):
    from transformer_mm_explainability_amd import visualbert_model as vm
    hidden, heads, inter, layers, vocab, max_pos, vdim, labels = (int(x) for x in g["dims"])
    model = vm.VisualBERT(vm.VisualBertConfig(hidden_size=hidden, num_attention_heads=heads, intermediate_size=inter,
                                              num_hidden_layers=layers, vocab_size=vocab, max_position_embeddings=max_pos,
                                              visual_embedding_dim=vdim, num_labels=labels))
    model.load_state_dict({k[3:]: t(v) for k, v in g.items() if k.startswith("w__")}, strict=False)
    sample = lambda: {"input_ids": t(g["input_ids"]).clone(), "input_mask": t(g["input_mask"]).clone(),       # noqa: E731
                      "segment_ids": torch.zeros_like(t(g["input_ids"])), "image_feature_0": t(g["image_feature_0"])}
    return model.eval(), sample


def test_visualbert_relprop_matches_reference_pass(golden, cpu_body):
    gm, g = golden("visualbert_model"), golden("visualbert_model_lrp")
    model, sample = visualbert_from_golden(gm)
    out = model(sample())["scores"]
    np.testing.assert_allclose(out.detach().numpy(), g["scores"], atol=1e-5)
    one_hot = torch.zeros_like(out)
    one_hot[0, int(g["index"])] = 1
    torch.sum(one_hot * out).backward()
    cam_in = model.relprop(one_hot.clone(), alpha=1, core=lrp_oracle.core)
    blocks = model.model.bert.encoder.layer
    within_reference_noise(torch.stack([b.attention.self.get_attn_cam() for b in blocks]), g, "attn_cam")
    within_reference_noise(cam_in, g, "cam_input")


def test_mask_add_rule_keeps_the_relevance_on_the_scores():
    """``Add.relprop`` of [scores, mask] with a mask that is zero wherever relevance arrives: the rule returns (to rounding)
    its input -- the property that makes LXMERT's skipped rule and VisualBERT's applied one agree on unpadded inputs."""
    g = torch.Generator().manual_seed(2)
    B, N, H, D = 2, 6, 3, 8
    q, k = torch.randn(B, N, H, D, generator=g), torch.randn(B, N, H, D, generator=g)
    mask = torch.zeros(B, 1, 1, N)
    mask[1, 0, 0, -2:] = -10000.0
    cam_p = torch.rand(B, H, N, N, generator=g)
    cam_p[1, :, :, -2:] = 0                                                # P is exactly 0 on masked keys
    out = bert_lrp._mask_add_relprop(cam_p, dict(q=q, k=k, mask=mask))
    torch.testing.assert_close(out, cam_p, rtol=1e-4, atol=1e-6)
