"""CPU suite: the LRP pass of the LXMERT / VisualBERT bodies (``bert_lrp.py`` + the ``relprop`` methods of ``lxmert_model`` /
``visualbert_model``) against the reference's REAL pass (``lxmert_lrp.py`` / ``BERT_ours.py`` over their LRP layer library;
fixtures ``lxmert_model_lrp.npz`` / ``visualbert_model_lrp.npz`` from ``tests/golden/make_golden.py``).

No GPU here: the capture op of the attention modules is replaced by a plain-torch stand-in (test infrastructure, below) and the
attention core of the pass runs on the referee ``bert_lrp.core_torch``; what is pinned is the host logic -- tapes, rule order,
Clone / Add / Linear closed forms, per-sample sums.  The HIP kernels are pinned on the same fixtures in ``tests/test_gpu_lrp.py``.

Tolerances: the pass divides by layer outputs (``safe_divide``) -- a near-zero denominator amplifies fp32 rounding of the two
implementations (closed form vs autograd-in-autograd) differently, so relevances are compared at 1e-4 of the tensor's max."""
import math
import types

import numpy as np
import pytest
import torch

from transformer_mm_explainability_amd import attention_modules, bert_lrp


def rel_close(got, want, rel=1e-4, what=""):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    bound = rel * max(float(np.abs(want).max()), 1e-30)
    err = float(np.abs(got - want).max())
    assert err <= bound, "%s: max |diff| %.3e > %.3e" % (what, err, bound)


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
    cam_lang, cam_vis = model.relprop(one_hot.clone(), alpha=1, core=bert_lrp.core_torch)
    for name, module in lxmert_cams(model).items():
        rel_close(module.get_attn_cam(), g["cam__" + name], what=name)
    rel_close(cam_lang, g["cam_lang"], what="cam_lang")
    rel_close(cam_vis, g["cam_vis"], what="cam_vis")
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
    cam_lang, _ = model.relprop(one_hot.clone(), alpha=1, core=bert_lrp.core_torch)
    rel_close(cam_lang[:1], g["cam_lang"], what="cam_lang of item 0 in a batch of 2")
    rel_close(model.lxmert.encoder.layer[0].attention.self.get_attn_cam()[:1], g["cam__l0"], what="l0 of item 0")


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
    cam_in = model.relprop(one_hot.clone(), alpha=1, core=bert_lrp.core_torch)
    blocks = model.model.bert.encoder.layer
    for i, b in enumerate(blocks):
        rel_close(b.attention.self.get_attn_cam(), g["attn_cam"][i], what="layer %d" % i)
    rel_close(cam_in, g["cam_input"], what="cam_input")


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
