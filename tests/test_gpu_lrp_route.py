"""-m gpu: the LRP route of the generators (minimal form).

``use_lrp=True`` is the DEFAULT of the reference's DETR / LXMERT ``generate_ours``.  The rule schedule is the same, only
the cam comes from ``get_attn_cam()`` (filled by the body's ``relprop``).  These tests plug a body that supplies
``attn_cam`` slabs + a ``relprop`` and compare every LRP-route entry point with what the REFERENCE's generator classes
returned on the same tensors (``tests/golden/*_chain_lrp.npz``, made by ``make_golden.py`` from the reference code).
A body without ``relprop`` must fail loudly, before any work is done.
"""
import types

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def close(a, b, atol=1e-5, rtol=0.0):
    """Relevancy maps are judged on the ABSOLUTE 1e-5 of the north star (``rtol = 0``; ``tests/parity.py`` records the largest
    error); logits / gradients / other intermediate tensors pass an explicit relative term."""
    import parity
    parity.close(a, b, atol=atol, rtol=rtol)


def cu(x):
    return torch.from_numpy(np.asarray(x)).cuda()


class SlotCam:
    def __init__(self, attn, grad, cam):
        self._a, self._g, self._c = attn, grad, cam

    def get_attn(self):
        return self._a

    def get_attn_gradients(self):
        return self._g

    def get_attn_cam(self):
        return self._c


class FakeBody(nn.Module):
    def __init__(self):
        super().__init__()
        self.dummy = nn.Parameter(torch.zeros(1))
        self.calls = []

    def relprop(self, one_hot, **kw):
        self.calls.append((one_hot.detach().clone(), dict(kw)))


def trios(g, prefix):
    return list(zip(cu(g[prefix + "_attn"]), cu(g[prefix + "_grad"]), cu(g[prefix + "_cam"])))


def detr_model(g):
    logits = cu(g["logits"]).requires_grad_(True)
    model = FakeBody()
    model.forward = lambda img: {"pred_logits": logits}
    enc, ds, dc = trios(g, "enc"), trios(g, "dself"), trios(g, "dcross")
    model.transformer = types.SimpleNamespace(
        encoder=types.SimpleNamespace(layers=[types.SimpleNamespace(self_attn=SlotCam(*t)) for t in enc]),
        decoder=types.SimpleNamespace(layers=[types.SimpleNamespace(self_attn=SlotCam(*ds[i]),
                                                                    multihead_attn=SlotCam(*dc[i]))
                                              for i in range(len(ds))]))
    return model


def test_detr_lrp_route(golden):
    from transformer_mm_explainability_amd import detr_explainability as de
    g = golden("detr_chain_lrp")
    tgt = cu(g["target_index"])
    model = detr_model(g)
    gen = de.Generator(model)
    out = gen.generate_ours(None, tgt)                       # the reference's default arguments (use_lrp=True)
    close(out, g["out_default"])
    close(gen.R_i_i, g["R_i_i"])
    close(gen.R_q_q, g["R_q_q"])
    one_hot, kw = model.calls[0]                             # relprop got the reference's arguments
    close(one_hot, g["relprop_one_hot"])
    assert kw["alpha"] == 1 and torch.equal(kw["target_index"], tgt)
    assert np.array_equal(kw["target_class"].cpu().numpy(), g["relprop_target_class"])
    close(de.Generator(detr_model(g)).generate_transformer_att(None, tgt), g["transformer_att_out"])
    close(de.Generator(detr_model(g)).generate_partial_lrp(None, tgt), g["partial_lrp_out"])
    close(de.GeneratorAlbationNoAgg(detr_model(g)).generate_ours_abl(None, tgt, use_lrp=True), g["abl_lrp_out"])
    # the ablation must forward apply_self_in_rule_10 (reference :342, :347-349)
    close(de.GeneratorAlbationNoAgg(detr_model(g)).generate_ours_abl(None, tgt, apply_self_in_rule_10=False),
          g["abl_noself_out"])


def lxmert_usage(g):
    lang, vis = trios(g, "lang"), trios(g, "vis")
    x = {k: trios(g, "x_" + k) for k in ("lang_cross", "img_cross", "lang_self", "img_self")}
    score = cu(g["score"]).requires_grad_(True)

    def sa(t):
        return types.SimpleNamespace(self=SlotCam(*t))

    model = FakeBody()
    model.device = torch.device("cuda")
    model.lxmert = types.SimpleNamespace(encoder=types.SimpleNamespace(
        layer=[types.SimpleNamespace(attention=sa(t)) for t in lang],
        r_layers=[types.SimpleNamespace(attention=sa(t)) for t in vis],
        x_layers=[types.SimpleNamespace(
            visual_attention=types.SimpleNamespace(att=SlotCam(*x["lang_cross"][i])),
            visual_attention_copy=types.SimpleNamespace(att=SlotCam(*x["img_cross"][i])),
            lang_self_att=sa(x["lang_self"][i]), visn_self_att=sa(x["img_self"][i])) for i in range(len(x["lang_self"]))]))
    T, I = g["lang_attn"].shape[-1], g["vis_attn"].shape[-1]
    return types.SimpleNamespace(model=model, text_len=T, image_boxes_len=I,
                                 forward=lambda item: types.SimpleNamespace(question_answering_score=score))


@pytest.mark.parametrize("fused", [True, False])
def test_lxmert_lrp_route(golden, fused):
    from transformer_mm_explainability_amd import lxmert_explainability as le
    g = golden("lxmert_chain_lrp")
    usage = lxmert_usage(g)
    gen = le.GeneratorOurs(usage)
    gen.fused = fused
    R_t_t, R_t_i = gen.generate_ours(None)                   # the reference's default arguments (use_lrp=True)
    close(R_t_t, g["R_t_t"])
    close(R_t_i, g["R_t_i"])
    close(gen.R_i_i, g["R_i_i"])
    close(gen.R_i_t, g["R_i_t"])
    close(usage.model.calls[0][0], g["relprop_one_hot"])
    assert usage.model.calls[0][1] == {"alpha": 1}
    r_tt, r_ti = le.GeneratorBaselines(lxmert_usage(g)).generate_transformer_attr(None)
    close(r_tt, g["tattr_R_t_t"])
    close(r_ti, g["tattr_R_t_i"])
    r_tt, r_ti = le.GeneratorBaselines(lxmert_usage(g)).generate_partial_lrp(None)
    close(r_tt, g["plrp_R_t_t"])
    close(r_ti, g["plrp_R_t_i"])


def test_visualbert_lrp_route(golden):
    from transformer_mm_explainability_amd import visualbert_explainability as vb
    g = golden("visualbert_chain_lrp")
    scores = cu(g["scores"]).requires_grad_(True)
    model = FakeBody()
    model.forward = lambda inp: {"scores": scores}
    model.model = types.SimpleNamespace(bert=types.SimpleNamespace(encoder=types.SimpleNamespace(
        layer=[types.SimpleNamespace(attention=types.SimpleNamespace(self=SlotCam(a, b, c)))
               for a, b, c in zip(cu(g["attn"]), cu(g["grad"]), cu(g["cam"]))])))
    inp = {"input_mask": cu(g["input_mask"])}
    close(vb.SelfAttentionGenerator(model).generate_transformer_att(inp), g["transformer_att_out"])
    close(vb.SelfAttentionGenerator(model).generate_transformer_att(inp, start_layer=1), g["transformer_att_s1"])
    close(vb.SelfAttentionGenerator(model).generate_partial_lrp(inp), g["partial_lrp_out"])
    close(model.calls[0][0], g["relprop_one_hot"])


def test_bodies_without_relprop_fail_before_any_work(golden):
    """A body WITHOUT an LRP pass (``detr_model`` has one since round 3, tests/test_gpu_lrp.py): the default-argument call
    names what is missing (and does not run a forward / backward first); the cam slot of the hooked modules is a plain
    slot an external LRP pass can fill."""
    from transformer_mm_explainability_amd import detr_explainability as de
    from transformer_mm_explainability_amd.attention_modules import MultiheadAttention

    class NoRelprop(nn.Module):
        def forward(self, img):
            raise AssertionError("must not run the body before checking for relprop")

    with pytest.raises(NotImplementedError, match="relprop"):
        de.Generator(NoRelprop()).generate_ours(None, torch.tensor([0]))
    with pytest.raises(NotImplementedError, match="relprop"):
        de.Generator(NoRelprop()).generate_transformer_att(None, torch.tensor([0]))
    with pytest.raises(NotImplementedError, match="relprop"):
        de.Generator(NoRelprop()).generate_partial_lrp(None, torch.tensor([0]))
    mha = MultiheadAttention(16, 2)
    with pytest.raises(NotImplementedError, match="save_attn_cam"):
        mha.get_attn_cam()
    cam = torch.zeros(2, 3, 3)
    mha.save_attn_cam(cam)
    assert mha.get_attn_cam() is cam
