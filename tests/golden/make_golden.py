#!/usr/bin/env python3
"""Generate golden input/output vectors by running the REFERENCE's own code in-process.

Run in the build container only (needs /root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

Writes ``tests/golden/*.npz``.  The fixtures pin (a) the numpy oracle (``oracle/``) and (b) the
HIP path (``-m gpu`` tests) to what the reference computes on identical seeded inputs.

What is executed from the reference (nothing is copied into this repo):
  * ``DETR/modules/ExplanationGenerator.py``           rule functions + ``Generator.generate_ours`` (+ baselines)
  * ``lxmert/lxmert/src/ExplanationGenerator.py``      rule functions + ``GeneratorOurs.generate_ours``
  * ``VisualBERT/.../backends/ExplanationGenerator.py`` ``compute_rollout_attention`` + ``SelfAttentionGenerator.generate_ours``
    (imported by file path with a stub ``cv2`` module: cv2 is only used for visualisation)
  * ``CLIP_explainability.ipynb`` cell 6 ``interpret``   exec'd from the notebook JSON
  * ``Transformer_MM_explainability_ViT.ipynb`` cell 7   exec'd from the notebook JSON
  * ``CLIP/clip/model.py`` + ``auxilary.py``             random-init tiny CLIP with the real hooks
  * ``DETR/modules/layers.py`` ``MultiheadAttention``    real ``save_attn`` / ``register_hook`` capture
``Tensor.cuda`` is patched to identity (the generators hard-code ``.cuda()``).
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = os.environ.get("MMX_REFERENCE", "/root/reference")
OUT = os.environ.get("MMX_GOLDEN_OUT") or os.path.dirname(os.path.abspath(__file__))   # tests regenerate into a scratch dir

torch.Tensor.cuda = lambda self, *a, **k: self  # CPU execution of hard-coded .cuda()
torch.set_num_threads(4)


def load_by_path(name, path, package=None):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


sys.path.insert(0, REF)
from DETR.modules import ExplanationGenerator as detr_eg  # noqa: E402
from DETR.modules import layers as detr_layers  # noqa: E402
from lxmert.lxmert.src import ExplanationGenerator as lx_eg  # noqa: E402

sys.modules.setdefault("cv2", types.ModuleType("cv2"))
vb_eg = load_by_path(
    "vb_eg", os.path.join(REF, "VisualBERT/mmf/models/transformers/backends/ExplanationGenerator.py"))


def softmax_attn(gen, *shape, causal=False):
    s = torch.randn(*shape, generator=gen)
    if causal:
        n = shape[-1]
        s = s + torch.full((n, n), float("-inf")).triu_(1)
    return s.softmax(-1)


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez(path, **{k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                      for k, v in arrays.items()})
    print("wrote", path, {k: tuple(v.shape) if hasattr(v, "shape") else () for k, v in arrays.items()
                           if not k.startswith("w__")})


# ----------------------------------------------------------------------------- rule-level fixtures
def gen_rules():
    g = torch.Generator().manual_seed(100)
    H, Ns, Nq = 4, 9, 13
    out = {}
    cam = softmax_attn(g, 1, H, Ns, Ns)
    grad = torch.randn(1, H, Ns, Ns, generator=g) * 0.05
    out["cam_ss"], out["grad_ss"] = cam, grad
    out["avg_heads_detr"] = detr_eg.avg_heads(cam, grad)
    out["avg_heads_lxmert"] = lx_eg.avg_heads(cam, grad)
    cam_bar = out["avg_heads_detr"]

    # relevancy state that satisfies handle_residual's diag >= 0 contract
    R_ss = torch.eye(Ns)
    R_sq = torch.rand(Ns, Nq, generator=g) * 0.1
    for _ in range(3):
        c = detr_eg.avg_heads(softmax_attn(g, H, Ns, Ns), torch.randn(H, Ns, Ns, generator=g) * 0.05)
        R_ss = R_ss + c @ R_ss
    R_qq = torch.eye(Nq)
    for _ in range(2):
        c = detr_eg.avg_heads(softmax_attn(g, H, Nq, Nq), torch.randn(H, Nq, Nq, generator=g) * 0.05)
        R_qq = R_qq + c @ R_qq
    R_qs = torch.rand(Nq, Ns, generator=g) * 0.1
    out.update(R_ss=R_ss, R_sq=R_sq, R_qq=R_qq, R_qs=R_qs)

    a, b = detr_eg.apply_self_attention_rules(R_ss, R_sq, cam_bar)
    out["self_rules_ss_add"], out["self_rules_sq_add"] = a, b
    out["handle_residual_ss"] = detr_eg.handle_residual(R_ss)
    out["handle_residual_qq_lx"] = lx_eg.handle_residual(R_qq)

    cam_sq = detr_eg.avg_heads(softmax_attn(g, H, Ns, Nq), torch.randn(H, Ns, Nq, generator=g) * 0.05)
    out["cam_sq"] = cam_sq
    out["mm_detr_norm"] = detr_eg.apply_mm_attention_rules(R_ss, R_qq, cam_sq.clone())
    out["mm_detr_nonorm"] = detr_eg.apply_mm_attention_rules(R_ss, R_qq, cam_sq.clone(), apply_normalization=False)
    out["mm_detr_noself"] = detr_eg.apply_mm_attention_rules(R_ss, R_qq, cam_sq.clone(), apply_self_in_rule_10=False)
    # NaN policy: identity R (0/0 rows in handle_residual) -> DETR zeroes, LXMERT propagates
    out["mm_detr_nan"] = detr_eg.apply_mm_attention_rules(torch.eye(Ns), torch.eye(Nq), cam_sq.clone())
    sq, ss = lx_eg.apply_mm_attention_rules(R_ss, R_qq, R_qs, cam_sq.clone())
    out["mm_lx_sq_add"], out["mm_lx_ss_add"] = sq, ss
    sq, ss = lx_eg.apply_mm_attention_rules(R_ss, R_qq, R_qs, cam_sq.clone(), apply_normalization=False)
    out["mm_lx_nonorm_sq_add"], out["mm_lx_nonorm_ss_add"] = sq, ss
    sq, ss = lx_eg.apply_mm_attention_rules(torch.eye(Ns), torch.eye(Nq), R_qs, cam_sq.clone())
    out["mm_lx_nan_sq_add"], out["mm_lx_nan_ss_add"] = sq, ss

    # rollout: DETR/LXMERT operate on [1,N,N]/[N,N] head-averaged maps, VisualBERT on [B,N,N]
    L = 4
    mats = [softmax_attn(g, 1, H, Ns, Ns).mean(1) for _ in range(L)]  # [1, N, N]
    out["rollout_in"] = torch.stack(mats)
    out["rollout_detr"] = detr_eg.compute_rollout_attention([m.clone() for m in mats])
    out["rollout_detr_s2"] = detr_eg.compute_rollout_attention([m.clone() for m in mats], start_layer=2)
    out["rollout_lxmert"] = lx_eg.compute_rollout_attention([m.clone() for m in mats])
    matsb = [softmax_attn(g, 3, H, Ns, Ns).mean(1) for _ in range(L)]  # [3, N, N]
    out["rollout_vb_in"] = torch.stack(matsb)
    out["rollout_vb"] = vb_eg.compute_rollout_attention([m.clone() for m in matsb])
    out["rollout_vb_s1"] = vb_eg.compute_rollout_attention([m.clone() for m in matsb], start_layer=1)

    cam4 = softmax_attn(g, 1, H, Nq, Ns)
    grad4 = torch.randn(1, H, Nq, Ns, generator=g)
    out["gradcam_cam"], out["gradcam_grad"] = cam4, grad4
    out["gradcam_out"] = detr_eg.Generator.gradcam(None, cam4, grad4)
    save("rules", **out)


# ----------------------------------------------------------------------------- fake models for the generators
class Slot:
    """Stands in for a hooked attention module: fixed attn / grad tensors."""

    def __init__(self, attn, grad):
        self._a, self._g = attn, grad

    def get_attn(self):
        return self._a

    def get_attn_gradients(self):
        return self._g

    def get_attn_cam(self):
        raise AssertionError("LRP cam is not on the fixtures' path (use_lrp=False)")

    # ViT naming (external ViT_new)
    get_attention_map = get_attn


class FakeBody(nn.Module):
    def __init__(self):
        super().__init__()
        self.dummy = nn.Parameter(torch.zeros(1))


def gen_detr_chain(tag, seed, H, Ni, Nq, Le, Ld, targets, **flags):
    g = torch.Generator().manual_seed(seed)
    enc = [(softmax_attn(g, H, Ni, Ni), torch.randn(H, Ni, Ni, generator=g) * 0.1) for _ in range(Le)]
    dself = [(softmax_attn(g, H, Nq, Nq), torch.randn(H, Nq, Nq, generator=g) * 0.1) for _ in range(Ld)]
    dcross = [(softmax_attn(g, H, Nq, Ni), torch.randn(H, Nq, Ni, generator=g) * 0.1) for _ in range(Ld)]
    n_cls = 7
    logits = torch.randn(1, Nq, n_cls, generator=g).requires_grad_(True)

    model = FakeBody()
    model.forward = lambda img: {"pred_logits": logits}
    model.transformer = types.SimpleNamespace(
        encoder=types.SimpleNamespace(layers=[types.SimpleNamespace(self_attn=Slot(a, gr)) for a, gr in enc]),
        decoder=types.SimpleNamespace(layers=[
            types.SimpleNamespace(self_attn=Slot(*dself[i]), multihead_attn=Slot(*dcross[i])) for i in range(Ld)]))
    gen = detr_eg.Generator(model)
    tgt = torch.tensor(targets)
    out = gen.generate_ours(None, tgt, use_lrp=False, **flags)
    arrays = dict(target_index=tgt, out=out,
                  enc_attn=torch.stack([a for a, _ in enc]), enc_grad=torch.stack([b for _, b in enc]),
                  dself_attn=torch.stack([a for a, _ in dself]), dself_grad=torch.stack([b for _, b in dself]),
                  dcross_attn=torch.stack([a for a, _ in dcross]), dcross_grad=torch.stack([b for _, b in dcross]),
                  R_i_i=gen.R_i_i, R_q_q=gen.R_q_q)
    if not flags:
        arrays["rollout_out"] = detr_eg.Generator(model).generate_rollout(None, tgt)
        arrays["raw_attn_out"] = detr_eg.Generator(model).generate_raw_attn(None, tgt)
        arrays["gradcam_out"] = detr_eg.Generator(model).generate_attn_gradcam(None, tgt)
        abl = detr_eg.GeneratorAlbationNoAgg(model)
        arrays["abl_out"] = abl.generate_ours_abl(None, tgt)
    save(tag, **arrays)


def gen_lxmert_chain(tag, seed, H, T, I, Ll, Lr, Lx, **flags):
    g = torch.Generator().manual_seed(seed)

    def pair(nq, nk):
        return softmax_attn(g, 1, H, nq, nk), torch.randn(1, H, nq, nk, generator=g) * 0.1

    lang = [pair(T, T) for _ in range(Ll)]
    vis = [pair(I, I) for _ in range(Lr)]
    xl = [dict(lang_cross=pair(T, I), img_cross=pair(I, T), lang_self=pair(T, T), img_self=pair(I, I))
          for _ in range(Lx)]
    score = torch.randn(1, 11, generator=g).requires_grad_(True)

    def sa(p):
        return types.SimpleNamespace(self=Slot(*p))

    model = FakeBody()
    model.device = torch.device("cpu")
    model.lxmert = types.SimpleNamespace(encoder=types.SimpleNamespace(
        layer=[types.SimpleNamespace(attention=sa(p)) for p in lang],
        r_layers=[types.SimpleNamespace(attention=sa(p)) for p in vis],
        x_layers=[types.SimpleNamespace(
            visual_attention=types.SimpleNamespace(att=Slot(*b["lang_cross"])),
            visual_attention_copy=types.SimpleNamespace(att=Slot(*b["img_cross"])),
            lang_self_att=sa(b["lang_self"]), visn_self_att=sa(b["img_self"])) for b in xl]))
    usage = types.SimpleNamespace(
        model=model, text_len=T, image_boxes_len=I,
        forward=lambda item: types.SimpleNamespace(question_answering_score=score))
    gen = lx_eg.GeneratorOurs(usage)
    R_t_t, R_t_i = gen.generate_ours(None, use_lrp=False, **flags)
    arrays = dict(R_t_t=R_t_t, R_t_i=R_t_i, R_i_i=gen.R_i_i, R_i_t=gen.R_i_t,
                  lang_attn=torch.stack([a for a, _ in lang]), lang_grad=torch.stack([b for _, b in lang]),
                  vis_attn=torch.stack([a for a, _ in vis]), vis_grad=torch.stack([b for _, b in vis]))
    for key in ("lang_cross", "img_cross", "lang_self", "img_self"):
        arrays["x_%s_attn" % key] = torch.stack([b[key][0] for b in xl])
        arrays["x_%s_grad" % key] = torch.stack([b[key][1] for b in xl])
    if not flags:
        abl = lx_eg.GeneratorOursAblationNoAggregation(usage)
        # default normalize_self_attention=True trips handle_residual's diag>=0 assert on R = cam@R
        a_tt, a_ti = abl.generate_ours_no_agg(None, use_lrp=False, normalize_self_attention=False)
        arrays["abl_R_t_t"], arrays["abl_R_t_i"] = a_tt, a_ti
        base = lx_eg.GeneratorBaselines(usage)
        r_tt, r_ti = base.generate_rollout(None)
        arrays["rollout_R_t_t"], arrays["rollout_R_t_i"] = r_tt, r_ti
        r_tt, r_ti = base.generate_raw_attn(None)
        arrays["raw_R_t_t"], arrays["raw_R_t_i"] = r_tt, r_ti
        r_tt, r_ti = base.generate_attn_gradcam(None)
        arrays["gradcam_R_t_t"], arrays["gradcam_R_t_i"] = r_tt, r_ti
    save(tag, **arrays)


def notebook_cell(nb_path, idx):
    nb = json.load(open(os.path.join(REF, nb_path)))
    return "".join(nb["cells"][idx]["source"])


def gen_vit_chain():
    g = torch.Generator().manual_seed(300)
    H, N, L = 3, 17, 4
    layers = [(softmax_attn(g, 1, H, N, N), torch.randn(1, H, N, N, generator=g) * 0.1) for _ in range(L)]
    logits = torch.randn(1, 10, generator=g).requires_grad_(True)
    model = FakeBody()
    model.forward = lambda x, register_hook=False: logits
    model.blocks = [types.SimpleNamespace(attn=Slot(a, gr)) for a, gr in layers]
    ns = {"torch": torch, "np": np}
    exec(notebook_cell("Transformer_MM_explainability_ViT.ipynb", 7), ns)
    out = ns["generate_relevance"](model, torch.zeros(1, 3, 8, 8), index=3)
    save("vit_chain", attn=torch.stack([a for a, _ in layers]), grad=torch.stack([b for _, b in layers]), out=out)


def gen_visualbert_chain():
    g = torch.Generator().manual_seed(400)
    H, N, L = 4, 21, 5
    layers = [(softmax_attn(g, 1, H, N, N), torch.randn(1, H, N, N, generator=g) * 0.1) for _ in range(L)]
    scores = torch.randn(1, 9, generator=g).requires_grad_(True)
    model = FakeBody()
    model.forward = lambda inp: {"scores": scores}
    model.model = types.SimpleNamespace(bert=types.SimpleNamespace(encoder=types.SimpleNamespace(
        layer=[types.SimpleNamespace(attention=types.SimpleNamespace(self=Slot(a, gr))) for a, gr in layers])))
    input_mask = torch.zeros(1, N, dtype=torch.long)
    input_mask[0, :8] = 1
    gen = vb_eg.SelfAttentionGenerator(model)
    out = gen.generate_ours({"input_mask": input_mask})
    roll = vb_eg.SelfAttentionGenerator(model).generate_rollout({"input_mask": input_mask})
    save("visualbert_chain", attn=torch.stack([a for a, _ in layers]), grad=torch.stack([b for _, b in layers]),
         input_mask=input_mask, out=out, rollout_out=roll)


# ----------------------------------------------------------------------------- LRP route on fake slots (round 2)
class SlotCam(Slot):
    """A hooked attention module after an LRP pass: ``get_attn_cam()`` returns a fixed per-head relevance."""

    def __init__(self, attn, grad, cam):
        super().__init__(attn, grad)
        self._c = cam

    def get_attn_cam(self):
        return self._c


def _lrp_cam(g, *shape):
    """Stand-in for an alpha-beta LRP relevance of the attention probabilities: signed, sparse-ish, O(1e-1)."""
    return torch.randn(*shape, generator=g) * 0.2 * (torch.rand(*shape, generator=g) > 0.3)


def gen_detr_chain_lrp():
    """``use_lrp=True`` (the DEFAULT of ``Generator.generate_ours``), ``generate_transformer_att`` and
    ``generate_partial_lrp`` of the reference on slots that carry an ``attn_cam``; ``model.relprop`` records its call."""
    g = torch.Generator().manual_seed(210)
    H, Ni, Nq, Le, Ld = 4, 23, 9, 2, 3

    def trio(nq, nk):
        return softmax_attn(g, H, nq, nk), torch.randn(H, nq, nk, generator=g) * 0.1, _lrp_cam(g, H, nq, nk)

    enc = [trio(Ni, Ni) for _ in range(Le)]
    dself = [trio(Nq, Nq) for _ in range(Ld)]
    dcross = [trio(Nq, Ni) for _ in range(Ld)]
    logits = torch.randn(1, Nq, 7, generator=g).requires_grad_(True)
    calls = []
    model = FakeBody()
    model.forward = lambda img: {"pred_logits": logits}
    model.relprop = lambda one_hot, **kw: calls.append((one_hot.detach().clone(), dict(kw)))
    model.transformer = types.SimpleNamespace(
        encoder=types.SimpleNamespace(layers=[types.SimpleNamespace(self_attn=SlotCam(*t)) for t in enc]),
        decoder=types.SimpleNamespace(layers=[
            types.SimpleNamespace(self_attn=SlotCam(*dself[i]), multihead_attn=SlotCam(*dcross[i])) for i in range(Ld)]))
    tgt = torch.tensor([5])
    gen = detr_eg.Generator(model)
    out = gen.generate_ours(None, tgt)                                   # default arguments: use_lrp=True
    assert len(calls) == 1 and calls[0][1]["alpha"] == 1
    arrays = dict(target_index=tgt, logits=logits, out_default=out, R_i_i=gen.R_i_i, R_q_q=gen.R_q_q,
                  relprop_one_hot=calls[0][0], relprop_target_class=calls[0][1]["target_class"],
                  transformer_att_out=detr_eg.Generator(model).generate_transformer_att(None, tgt),
                  partial_lrp_out=detr_eg.Generator(model).generate_partial_lrp(None, tgt),
                  abl_lrp_out=detr_eg.GeneratorAlbationNoAgg(model).generate_ours_abl(None, tgt, use_lrp=True),
                  abl_noself_out=detr_eg.GeneratorAlbationNoAgg(model).generate_ours_abl(
                      None, tgt, apply_self_in_rule_10=False))
    for name, layers in (("enc", enc), ("dself", dself), ("dcross", dcross)):
        arrays[name + "_attn"] = torch.stack([t[0] for t in layers])
        arrays[name + "_grad"] = torch.stack([t[1] for t in layers])
        arrays[name + "_cam"] = torch.stack([t[2] for t in layers])
    save("detr_chain_lrp", **arrays)


def gen_lxmert_chain_lrp():
    g = torch.Generator().manual_seed(260)
    H, T, I, Ll, Lr, Lx = 4, 8, 11, 3, 2, 3

    def trio(nq, nk):
        return (softmax_attn(g, 1, H, nq, nk), torch.randn(1, H, nq, nk, generator=g) * 0.1,
                _lrp_cam(g, 1, H, nq, nk).abs())          # non-negative cams keep handle_residual's diag >= 0 contract

    lang = [trio(T, T) for _ in range(Ll)]
    vis = [trio(I, I) for _ in range(Lr)]
    xl = [dict(lang_cross=trio(T, I), img_cross=trio(I, T), lang_self=trio(T, T), img_self=trio(I, I))
          for _ in range(Lx)]
    score = torch.randn(1, 11, generator=g).requires_grad_(True)
    calls = []

    def sa(t):
        return types.SimpleNamespace(self=SlotCam(*t))

    model = FakeBody()
    model.device = torch.device("cpu")
    model.relprop = lambda one_hot, **kw: calls.append((one_hot.detach().clone(), dict(kw)))
    model.lxmert = types.SimpleNamespace(encoder=types.SimpleNamespace(
        layer=[types.SimpleNamespace(attention=sa(t)) for t in lang],
        r_layers=[types.SimpleNamespace(attention=sa(t)) for t in vis],
        x_layers=[types.SimpleNamespace(
            visual_attention=types.SimpleNamespace(att=SlotCam(*b["lang_cross"])),
            visual_attention_copy=types.SimpleNamespace(att=SlotCam(*b["img_cross"])),
            lang_self_att=sa(b["lang_self"]), visn_self_att=sa(b["img_self"])) for b in xl]))
    usage = types.SimpleNamespace(model=model, text_len=T, image_boxes_len=I,
                                  forward=lambda item: types.SimpleNamespace(question_answering_score=score))
    gen = lx_eg.GeneratorOurs(usage)
    R_t_t, R_t_i = gen.generate_ours(None)                               # default arguments: use_lrp=True
    assert len(calls) == 1
    arrays = dict(score=score, R_t_t=R_t_t, R_t_i=R_t_i, R_i_i=gen.R_i_i, R_i_t=gen.R_i_t, relprop_one_hot=calls[0][0])
    base = lx_eg.GeneratorBaselines(usage)
    arrays["tattr_R_t_t"], arrays["tattr_R_t_i"] = base.generate_transformer_attr(None)
    arrays["plrp_R_t_t"], arrays["plrp_R_t_i"] = lx_eg.GeneratorBaselines(usage).generate_partial_lrp(None)
    for name, layers in (("lang", lang), ("vis", vis)):
        for j, part in enumerate(("attn", "grad", "cam")):
            arrays["%s_%s" % (name, part)] = torch.stack([t[j] for t in layers])
    for key in ("lang_cross", "img_cross", "lang_self", "img_self"):
        for j, part in enumerate(("attn", "grad", "cam")):
            arrays["x_%s_%s" % (key, part)] = torch.stack([b[key][j] for b in xl])
    save("lxmert_chain_lrp", **arrays)


def gen_visualbert_chain_lrp():
    g = torch.Generator().manual_seed(410)
    H, N, L = 4, 19, 4
    layers = [(softmax_attn(g, 1, H, N, N), torch.randn(1, H, N, N, generator=g) * 0.1, _lrp_cam(g, 1, H, N, N))
              for _ in range(L)]
    scores = torch.randn(1, 9, generator=g).requires_grad_(True)
    calls = []
    model = FakeBody()
    model.forward = lambda inp: {"scores": scores}
    model.relprop = lambda one_hot, **kw: calls.append((one_hot.detach().clone(), dict(kw)))
    model.model = types.SimpleNamespace(bert=types.SimpleNamespace(encoder=types.SimpleNamespace(
        layer=[types.SimpleNamespace(attention=types.SimpleNamespace(self=SlotCam(*t))) for t in layers])))
    input_mask = torch.zeros(1, N, dtype=torch.long)
    input_mask[0, :7] = 1
    arrays = dict(scores=scores, input_mask=input_mask,
                  attn=torch.stack([t[0] for t in layers]), grad=torch.stack([t[1] for t in layers]),
                  cam=torch.stack([t[2] for t in layers]),
                  transformer_att_out=vb_eg.SelfAttentionGenerator(model).generate_transformer_att({"input_mask": input_mask}),
                  transformer_att_s1=vb_eg.SelfAttentionGenerator(model).generate_transformer_att(
                      {"input_mask": input_mask}, start_layer=1),
                  partial_lrp_out=vb_eg.SelfAttentionGenerator(model).generate_partial_lrp({"input_mask": input_mask}))
    assert len(calls) == 3
    arrays["relprop_one_hot"] = calls[0][0]
    save("visualbert_chain_lrp", **arrays)


# ----------------------------------------------------------------------------- real hooked modules
def _clip_tiny_reference():
    """The reference's CLIP class (CLIP/clip/model.py) at a tiny configuration + the inputs of the clip_tiny fixtures."""
    load_by_path("clip_ref_pkg.auxilary", os.path.join(REF, "CLIP/clip/auxilary.py"))
    pkg = types.ModuleType("clip_ref_pkg")
    pkg.__path__ = [os.path.join(REF, "CLIP/clip")]
    sys.modules["clip_ref_pkg"] = pkg
    spec = importlib.util.spec_from_file_location("clip_ref_pkg.model", os.path.join(REF, "CLIP/clip/model.py"))
    model_mod = importlib.util.module_from_spec(spec)
    model_mod.__package__ = "clip_ref_pkg"
    sys.modules["clip_ref_pkg.model"] = model_mod
    spec.loader.exec_module(model_mod)

    torch.manual_seed(0)
    cfg = dict(embed_dim=32, image_resolution=32, vision_layers=3, vision_width=128, vision_patch_size=8,
               context_length=12, vocab_size=64, transformer_width=64, transformer_heads=2, transformer_layers=3)
    model = model_mod.CLIP(**cfg).float().eval()
    B = 3
    g = torch.Generator().manual_seed(1)
    image = torch.randn(1, 3, 32, 32, generator=g)
    texts = torch.zeros(B, cfg["context_length"], dtype=torch.long)
    g2 = torch.Generator().manual_seed(2)
    for b in range(B):
        n = 3 + b * 2
        texts[b, 0] = cfg["vocab_size"] - 2
        texts[b, 1:1 + n] = torch.randint(1, cfg["vocab_size"] - 2, (n,), generator=g2)
        texts[b, 1 + n] = cfg["vocab_size"] - 1  # EOT = arg-max id
    return model_mod, model, cfg, image, texts


def gen_clip_tiny():
    model_mod, model, cfg, image, texts = _clip_tiny_reference()
    B = texts.shape[0]

    ns = {"torch": torch, "np": np, "start_layer": -1, "start_layer_text": -1}
    exec(notebook_cell("CLIP_explainability.ipynb", 6), ns)
    interpret = ns["interpret"]
    arrays = {"image": image, "texts": texts, "cfg_json": json.dumps(cfg)}
    for k, v in model.state_dict().items():
        arrays["w__" + k] = v
    for tag, sl, slt in (("last", -1, -1), ("all", 0, 0), ("mid", 1, 2)):
        R_text, R_image = interpret(image, texts, model, "cpu", start_layer=sl, start_layer_text=slt)
        arrays["R_text_" + tag], arrays["R_image_" + tag] = R_text, R_image
    # captured buffers of the last forward + ONE backward through every hook
    logits_per_image, _ = model(image.repeat(B, 1, 1, 1), texts)
    arrays["logits_per_image"] = logits_per_image
    one_hot = torch.sum(torch.eye(B) * logits_per_image)
    model.zero_grad()
    one_hot.backward()
    vis_blocks = list(model.visual.transformer.resblocks.children())
    txt_blocks = list(model.transformer.resblocks.children())
    arrays["img_attn"] = torch.stack([b.attn_probs for b in vis_blocks])
    arrays["img_grad"] = torch.stack([b.attn_grad for b in vis_blocks])
    arrays["txt_attn"] = torch.stack([b.attn_probs for b in txt_blocks])
    arrays["txt_grad"] = torch.stack([b.attn_grad for b in txt_blocks])
    save("clip_tiny", **arrays)


def gen_clip_tiny_fp16():
    """The reference's OWN half-precision mode: ``convert_weights`` (CLIP/clip/model.py:381-402: Linear / Conv / attention / projection
    parameters to fp16; LayerNorm computes in fp32 and casts back, model.py:157-164) on the clip_tiny model -- same seed, so the fp16
    weights are the fp32 ones of clip_tiny.npz rounded -- and notebook cell 6 unchanged: R is created in the dtype of the attention
    probabilities (cell 6:20,43), so the whole chain runs in fp16.  Run on the CPU (fp16 GEMMs accumulate in fp32 there, as on a GPU)."""
    model_mod, model, cfg, image, texts = _clip_tiny_reference()
    model_mod.convert_weights(model)
    assert model.dtype == torch.float16
    B = texts.shape[0]
    ns = {"torch": torch, "np": np, "start_layer": -1, "start_layer_text": -1}
    exec(notebook_cell("CLIP_explainability.ipynb", 6), ns)
    arrays = {}
    for tag, sl, slt in (("last", -1, -1), ("all", 0, 0), ("mid", 1, 2)):
        R_text, R_image = ns["interpret"](image, texts, model, "cpu", start_layer=sl, start_layer_text=slt)
        assert R_text.dtype == torch.float16 and R_image.dtype == torch.float16
        arrays["R_text_" + tag], arrays["R_image_" + tag] = R_text, R_image
    logits_per_image, _ = model(image.repeat(B, 1, 1, 1), texts)
    arrays["logits_per_image"] = logits_per_image
    one_hot = torch.sum(torch.eye(B) * logits_per_image)
    model.zero_grad()
    one_hot.backward()
    vis_blocks = list(model.visual.transformer.resblocks.children())
    txt_blocks = list(model.transformer.resblocks.children())
    arrays["img_attn"] = torch.stack([b.attn_probs for b in vis_blocks])
    arrays["img_grad"] = torch.stack([b.attn_grad for b in vis_blocks])
    arrays["txt_attn"] = torch.stack([b.attn_probs for b in txt_blocks])
    arrays["txt_grad"] = torch.stack([b.attn_grad for b in txt_blocks])
    save("clip_tiny_fp16", **arrays)


def gen_clip_vitb32_fp16():
    """The reference's model at the HEADLINE geometry (CLIP ViT-B/32, 12 + 12 layers) after ``convert_weights``, notebook cell 6 on
    a batch of 8 captions (bench.py's synthetic inputs), all layers -- fp16 and, for scale, fp32.  The weights are not stored: they
    are ``clip_model.random_init("ViT-B/32", seed=0)`` (the product's initialiser is only a source of numbers here; the state dict
    loads into the reference's class unchanged)."""
    model_mod, _, _, _, _ = _clip_tiny_reference()
    sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")))
    from transformer_mm_explainability_amd import clip_model
    sd = clip_model.random_init("ViT-B/32", seed=0).state_dict()
    cfg = dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768, vision_patch_size=32,
               context_length=77, vocab_size=49408, transformer_width=512, transformer_heads=8, transformer_layers=12)
    B = 8
    g = torch.Generator().manual_seed(1)                                   # tests/test_gpu_parity_fullsize.bench_inputs(8)
    image = torch.randn(1, 3, 224, 224, generator=g)
    texts = torch.zeros(B, 77, dtype=torch.long)
    g2 = torch.Generator().manual_seed(2)
    for b in range(B):
        n = int(torch.randint(3, 11, (1,), generator=g2))
        texts[b, 0] = 49406
        texts[b, 1:1 + n] = torch.randint(1, 49405, (n,), generator=g2)
        texts[b, 1 + n] = 49407
    ns = {"torch": torch, "np": np, "start_layer": -1, "start_layer_text": -1}
    exec(notebook_cell("CLIP_explainability.ipynb", 6), ns)
    arrays = {"texts": texts}
    for tag in ("fp32", "fp16"):
        model = model_mod.CLIP(**cfg).float().eval()
        model.load_state_dict(sd)
        if tag == "fp16":
            model_mod.convert_weights(model)
        R_text, R_image = ns["interpret"](image, texts, model, "cpu", start_layer=0, start_layer_text=0)
        arrays["R_text_" + tag], arrays["R_image_" + tag] = R_text, R_image
    save("clip_vitb32_fp16", **arrays)


def gen_detr_mha():
    torch.manual_seed(5)
    E, H, T, S, B = 64, 4, 6, 15, 2
    mha = detr_layers.MultiheadAttention(E, H, dropout=0.0).eval()
    g = torch.Generator().manual_seed(6)
    q = torch.randn(T, B, E, generator=g, requires_grad=True)
    k = torch.randn(S, B, E, generator=g, requires_grad=True)
    v = torch.randn(S, B, E, generator=g, requires_grad=True)
    out = mha(q, k, v)
    up = torch.randn(T, B, E, generator=g)
    (out * up).sum().backward()
    arrays = dict(query=q, key=k, value=v, upstream=up, out=out, attn=mha.get_attn(),
                  attn_grad=mha.get_attn_gradients(), dquery=q.grad, dkey=k.grad, dvalue=v.grad,
                  num_heads=np.int64(H))
    for name, p in mha.state_dict().items():
        arrays["w__" + name] = p
    save("detr_mha", **arrays)


def gen_detr_transformer():
    """The REAL reference transformer body (DETR/models/transformer.py, hooked MHA from DETR/modules/layers.py) with the
    heads of DETR/models/detr.py:34-38 on a synthetic backbone feature map, driven by the reference Generator."""
    detr_tr = load_by_path("detr_transformer_ref", os.path.join(REF, "DETR/models/transformer.py"))
    torch.manual_seed(11)
    d, heads, Le, Ld, ff, Q, n_cls, Cb, h, w = 32, 4, 2, 3, 64, 7, 5, 24, 3, 5

    class Body(nn.Module):
        def __init__(self):
            super().__init__()
            self.transformer = detr_tr.Transformer(d_model=d, nhead=heads, num_encoder_layers=Le,
                                                   num_decoder_layers=Ld, dim_feedforward=ff, dropout=0.0,
                                                   return_intermediate_dec=True)
            self.class_embed = nn.Linear(d, n_cls + 1)
            self.query_embed = nn.Embedding(Q, d)
            self.input_proj = nn.Conv2d(Cb, d, kernel_size=1)

        def forward(self, feats):
            mask = torch.zeros(feats.shape[0], h, w, dtype=torch.bool)
            hs, _ = self.transformer(self.input_proj(feats), mask, self.query_embed.weight, self.pos)
            return {"pred_logits": self.class_embed(hs)[-1]}

    body = Body().eval()
    g = torch.Generator().manual_seed(12)
    feats = torch.randn(1, Cb, h, w, generator=g)
    body.pos = torch.randn(1, d, h, w, generator=g)
    tgt = torch.tensor([1, 4])
    gen = detr_eg.Generator(body)
    out = gen.generate_ours(feats, tgt, use_lrp=False)
    arrays = dict(features=feats, pos=body.pos, target_index=tgt, out=out, R_i_i=gen.R_i_i, R_q_q=gen.R_q_q,
                  pred_logits=body(feats)["pred_logits"],
                  rollout_out=detr_eg.Generator(body).generate_rollout(feats, tgt),
                  raw_attn_out=detr_eg.Generator(body).generate_raw_attn(feats, tgt),
                  dims=np.array([d, heads, Le, Ld, ff, Q, n_cls, Cb, h, w]))
    for name, p in body.state_dict().items():
        arrays["w__" + name] = p
    # sine position embedding (DETR/models/position_encoding.py:12-48) on a mask with right/bottom padding
    misc = types.ModuleType("DETR.util.misc")
    misc.NestedTensor = object
    sys.modules.setdefault("DETR.util.misc", misc)
    pe = load_by_path("detr_pos_ref", os.path.join(REF, "DETR/models/position_encoding.py"))
    pad = torch.zeros(2, 4, 6, dtype=torch.bool)
    pad[1, 3:, :] = True
    pad[1, :, 4:] = True
    nested = types.SimpleNamespace(tensors=torch.zeros(2, 1, 4, 6), mask=pad)
    arrays["sine_mask"] = pad
    arrays["sine_pos"] = pe.PositionEmbeddingSine(d // 2, normalize=True)(nested)
    arrays["sine_pos_raw"] = pe.PositionEmbeddingSine(d // 2, normalize=False)(nested)
    save("detr_transformer", **arrays)


def gen_detr_transformer_lrp():
    """The reference's REAL LRP pass: ``DETR/models/transformer.py`` (every ``relprop`` of the
    encoder / decoder stack) over the LRP layer library ``DETR/modules/layers.py`` (``Linear`` / ``Add`` / ``Clone`` /
    ``einsum`` / ``IndexSelect`` / ``MultiheadAttention.relprop``, :770-801), entered through ``DETR.relprop``
    (``DETR/models/detr.py:79-92``, its source exec'd unchanged: ``detr.py`` itself needs torchvision to import).
    ``DETR.forward`` after the backbone (``detr.py:61-70``) is restated below (flagged glue): input projection,
    transformer, ``class_embed`` (an LRP ``Linear``), ``index_select`` of the LAST decoder level (the reference hard-codes
    level 5 of 6).  Recorded: every attention module's ``attn_cam`` after ``Generator.generate_ours(img, t)`` with its
    DEFAULT arguments (``use_lrp=True``), the outputs of the three LRP methods, and the relevance ``relprop`` returns."""
    detr_tr = load_by_path("detr_transformer_ref", os.path.join(REF, "DETR/models/transformer.py"))
    torch.manual_seed(21)
    d, heads, Le, Ld, ff, Q, n_cls, Cb, h, w = 32, 4, 2, 3, 64, 7, 5, 24, 3, 5

    class Body(nn.Module):
        def __init__(self):
            super().__init__()
            self.transformer = detr_tr.Transformer(d_model=d, nhead=heads, num_encoder_layers=Le,
                                                   num_decoder_layers=Ld, dim_feedforward=ff, dropout=0.0,
                                                   return_intermediate_dec=True)
            self.class_embed = detr_layers.Linear(d, n_cls + 1)
            self.query_embed = nn.Embedding(Q, d)
            self.input_proj = nn.Conv2d(Cb, d, kernel_size=1)
            self.index_select = detr_layers.IndexSelect()

        def forward(self, feats):                                     # detr.py:61-70 after the backbone (restated glue)
            mask = torch.zeros(feats.shape[0], h, w, dtype=torch.bool)
            hs, memory = self.transformer(self.input_proj(feats), mask, self.query_embed.weight, self.pos)
            self.memory_shape = memory.shape
            outputs_class = self.class_embed(hs)
            a = self.index_select(outputs_class, 0, torch.tensor([Ld - 1])).squeeze(0)
            return {"pred_logits": a}

    Body.relprop = reference_function("DETR/models/detr.py", "DETR", "relprop", {"torch": torch})
    body = Body().eval()
    g = torch.Generator().manual_seed(22)
    feats = torch.randn(1, Cb, h, w, generator=g)
    body.pos = torch.randn(1, d, h, w, generator=g)
    tgt = torch.tensor([1, 4])

    def record(feats):
        gen = detr_eg.Generator(body)
        out = gen.generate_ours(feats, tgt)                               # DEFAULT arguments: use_lrp=True
        enc, dec = body.transformer.encoder.layers, body.transformer.decoder.layers
        arrays = dict(out_default=out, R_i_i=gen.R_i_i, R_q_q=gen.R_q_q,
                      pred_logits=body(feats)["pred_logits"],
                      enc_cam=torch.stack([b.self_attn.get_attn_cam() for b in enc]),
                      dself_cam=torch.stack([b.self_attn.get_attn_cam() for b in dec]),
                      dcross_cam=torch.stack([b.multihead_attn.get_attn_cam() for b in dec]),
                      enc_attn=torch.stack([b.self_attn.get_attn() for b in enc]),
                      enc_grad=torch.stack([b.self_attn.get_attn_gradients() for b in enc]))
        # the relevance the pass hands back for the transformer input (conservation check + end-to-end pin of every relprop)
        outputs = body(feats)["pred_logits"]
        index = outputs[0, tgt, :-1].max(1)[1]
        one_hot = torch.zeros_like(outputs)
        one_hot[0, tgt, index] = 1
        body.zero_grad()
        torch.sum(one_hot * outputs).backward(retain_graph=True)
        arrays["cam_src"] = body.relprop(one_hot.clone(), alpha=1, target_index=tgt, target_class=index)
        arrays["target_class"] = index
        arrays["transformer_att_out"] = detr_eg.Generator(body).generate_transformer_att(feats, tgt)
        arrays["partial_lrp_out"] = detr_eg.Generator(body).generate_partial_lrp(feats, tgt)
        arrays["abl_lrp_out"] = detr_eg.GeneratorAlbationNoAgg(body).generate_ours_abl(feats, tgt, use_lrp=True)
        single = torch.tensor([4])
        arrays["out_default_single"] = detr_eg.Generator(body).generate_ours(feats, single)
        return {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in arrays.items()}

    arrays = dict(features=feats, pos=body.pos, target_index=tgt, dims=np.array([d, heads, Le, Ld, ff, Q, n_cls, Cb, h, w]))
    arrays.update(record(feats))
    for name, p in body.state_dict().items():
        arrays["w__" + name] = p.detach().clone()
    # Round 6: the same reference pass in float64 ("f64__" entries), as for LXMERT / VisualBERT: the distance between the reference's
    # OWN fp32 and fp64 passes is the yardstick of the GPU tests for every tensor that went through safe_divide
    torch.set_default_dtype(torch.float64)
    try:
        body.double()
        body.pos = body.pos.double()
        arrays.update({"f64__" + k: v for k, v in record(feats.double()).items() if k != "target_class"})
    finally:
        torch.set_default_dtype(torch.float32)
    save("detr_transformer_lrp", **arrays)



def gen_lrp_layers():
    """The LRP layer library itself (DETR/modules/layers.py): ``Linear`` / ``Add`` / ``Clone`` / ``IndexSelect`` relprops and the
    whole ``MultiheadAttention.relprop`` (:770-801) on seeded inputs -- pins ``lrp.py``'s closed forms (CPU suite) and the HIP
    attention-core kernels (GPU suite).  Two MHA cases: a generic one, and one whose value stream is all zeros (decoder
    layer 0 of DETR: ``tgt = 0``), which takes the q / k rescale branch (:791-799)."""
    L = detr_layers
    g = torch.Generator().manual_seed(31)
    arrays = {}
    torch.manual_seed(30)                                                # L.Linear draws its weights from the GLOBAL generator
    lin = L.Linear(12, 7)
    x = torch.randn(5, 3, 12, generator=g)
    lin(x)
    r = torch.randn(5, 3, 7, generator=g)
    arrays.update(lin_w=lin.weight, lin_x=x, lin_r=r, lin_out=lin.relprop(r, 1))
    add = L.Add()
    a, b = torch.randn(4, 6, generator=g), torch.randn(4, 6, generator=g)
    add([a, b])
    r = torch.randn(4, 6, generator=g)
    ra, rb = add.relprop(r, 1)
    arrays.update(add_a=a, add_b=b, add_r=r, add_out_a=ra, add_out_b=rb)
    clone = L.Clone()
    x = torch.randn(4, 6, generator=g)
    x[0, 0] = 0.0                                                        # safe_divide's zero branch
    clone(x, 3)
    rs = [torch.randn(4, 6, generator=g) for _ in range(3)]
    arrays.update(clone_x=x, clone_r=torch.stack(rs), clone_out=clone.relprop(rs, 1))
    sel = L.IndexSelect()
    x = torch.randn(3, 2, 5, generator=g)
    idx = torch.tensor([2])
    sel(x, 0, idx)
    r = torch.randn(1, 2, 5, generator=g)
    arrays.update(sel_x=x, sel_r=r, sel_out=sel.relprop(r, 1))
    for tag, zero_value, (E, H, T, S, B) in (("mha", False, (64, 4, 6, 15, 2)), ("mha0", True, (32, 2, 5, 5, 1))):
        torch.manual_seed(32 + zero_value)
        mha = L.MultiheadAttention(E, H, dropout=0.0).eval()
        q = torch.randn(T, B, E, generator=g)
        k = torch.randn(S, B, E, generator=g)
        v = torch.zeros(S, B, E) if zero_value else torch.randn(S, B, E, generator=g)
        out = mha(q, k, v)
        cam_out = torch.randn(T, B, E, generator=g) * 0.1
        cam_q, cam_k, cam_v = mha.relprop(cam_out, 1)
        arrays.update({tag + "_query": q, tag + "_key": k, tag + "_value": v, tag + "_out": out, tag + "_cam_out": cam_out,
                       tag + "_cam_q": cam_q, tag + "_cam_k": cam_k, tag + "_cam_v": cam_v,
                       tag + "_attn_cam": mha.get_attn_cam(), tag + "_attn": mha.get_attn(),
                       tag + "_heads": np.int64(H)})
        for name, p_ in mha.state_dict().items():
            arrays[tag + "_w__" + name] = p_
    save("lrp_layers", **arrays)


def _build_lxmert_ref():
    """The REAL reference LXMERT body (lxmert/lxmert/src/lxmert_lrp.py: embeddings, encoder with hooked attention,
    pooler, answer head) driven by the reference GeneratorOurs / GeneratorBaselines.  ``LxmertModel`` itself derives
    from a transformers base class whose API moved, so its forward (mask extension, lxmert_lrp.py:1188-1225) is
    restated in the wrapper; every layer that computes is the reference's own."""
    from transformers.models.lxmert.configuration_lxmert import LxmertConfig
    import transformers.file_utils as fu
    fu.add_code_sample_docstrings = lambda *a, **k: (lambda f: f)
    shim = types.ModuleType("transformers.configuration_lxmert")
    shim.LxmertConfig = LxmertConfig
    sys.modules["transformers.configuration_lxmert"] = shim
    from lxmert.lxmert.src import lxmert_lrp as lrp

    torch.manual_seed(21)
    T, I = 9, 11
    cfg = LxmertConfig(hidden_size=48, num_attention_heads=4, intermediate_size=96, l_layers=3, x_layers=3,
                       r_layers=2, visual_feat_dim=20, visual_pos_dim=4, vocab_size=60, num_qa_labels=13,
                       max_position_embeddings=32, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)

    class RefModel(nn.Module):
        def __init__(self):
            super().__init__()
            self.embeddings = lrp.LxmertEmbeddings(cfg)
            self.encoder = lrp.LxmertEncoder(cfg)
            self.pooler = lrp.LxmertPooler(cfg)

        def forward(self, input_ids, visual_feats, visual_pos, attention_mask, token_type_ids):
            ext = (1.0 - attention_mask[:, None, None, :].float()) * -10000.0
            emb = self.embeddings(input_ids, token_type_ids, None)
            vis_out, lang_out = self.encoder(emb, ext, visual_feats=visual_feats, visual_pos=visual_pos,
                                             visual_attention_mask=None, output_attentions=False)[:2]
            return self.pooler(lang_out[0][-1])

    class RefQA(nn.Module):
        def __init__(self):
            super().__init__()
            self.lxmert = RefModel()
            self.answer_head = lrp.LxmertVisualAnswerHead(cfg, cfg.num_qa_labels)
            self.device = torch.device("cpu")

        def forward(self, **kw):
            return types.SimpleNamespace(question_answering_score=self.answer_head(self.lxmert(**kw)))

    model = RefQA().eval()
    # LayerNorm affine / embedding rows away from their init so every parameter matters
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
    weights = {k: v.clone() for k, v in model.state_dict().items()}        # before the lazy deepcopy adds keys
    g = torch.Generator().manual_seed(22)
    inputs = dict(input_ids=torch.randint(1, 60, (1, T), generator=g),
                  visual_feats=torch.randn(1, I, 20, generator=g), visual_pos=torch.rand(1, I, 4, generator=g),
                  attention_mask=torch.ones(1, T), token_type_ids=torch.zeros(1, T, dtype=torch.long))
    # no padded text tokens: a padded query row has zero gradient, so its row of R_tt - I sums to 0 and the reference's
    # handle_residual (no nan_to_num in the LXMERT flavour) trips its own ``assert diag.min() >= 0`` on the NaN
    return model, weights, inputs, cfg, T, I


def gen_lxmert_model():
    model, weights, inputs, cfg, T, I = _build_lxmert_ref()
    usage = types.SimpleNamespace(model=model, text_len=T, image_boxes_len=I, forward=lambda item: model(**inputs))
    gen = lx_eg.GeneratorOurs(usage)
    R_t_t, R_t_i = gen.generate_ours(None, use_lrp=False)
    arrays = dict(R_t_t=R_t_t, R_t_i=R_t_i, R_i_i=gen.R_i_i, R_i_t=gen.R_i_t,
                  score=model(**inputs).question_answering_score,
                  dims=np.array([cfg.hidden_size, cfg.num_attention_heads, cfg.intermediate_size, cfg.l_layers,
                                 cfg.x_layers, cfg.r_layers, cfg.visual_feat_dim, cfg.vocab_size, cfg.num_qa_labels,
                                 cfg.max_position_embeddings, T, I]))
    base = lx_eg.GeneratorBaselines(usage)
    arrays["rollout_R_t_t"], arrays["rollout_R_t_i"] = base.generate_rollout(None)
    arrays["raw_R_t_t"], arrays["raw_R_t_i"] = base.generate_raw_attn(None)
    arrays["gradcam_R_t_t"], arrays["gradcam_R_t_i"] = base.generate_attn_gradcam(None)
    for k, v in inputs.items():
        arrays["in__" + k] = v
    for k, v in weights.items():
        arrays["w__" + k] = v
    save("lxmert_model", **arrays)


def _build_visualbert_ref():
    """The REAL reference BERT stack of VisualBERT (backends/BERT_ours.py: BertEncoder with hooked BertSelfAttention,
    BertPredictionHeadTransform) driven by the reference SelfAttentionGenerator.  mmf's own modules (the
    visio-linguistic embedding sum, the VisualBERT wrapper's sample_list massaging) need an installed ``mmf`` to
    import, so those few lines -- embeddings.py:325-460, visual_bert.py:340-395 and :568-600 -- are restated in the
    wrapper; all layers with weights in the encoder / head are the reference's classes."""
    import importlib
    from transformers import BertConfig
    pkg = types.ModuleType("vb_backends")
    pkg.__path__ = [os.path.join(REF, "VisualBERT/mmf/models/transformers/backends")]
    sys.modules["vb_backends"] = pkg
    bo = importlib.import_module("vb_backends.BERT_ours")

    torch.manual_seed(31)
    T, V, Tpad, vdim, labels = 8, 10, 12, 20, 9
    cfg = BertConfig(hidden_size=48, num_attention_heads=4, intermediate_size=96, num_hidden_layers=4, vocab_size=70,
                     max_position_embeddings=32, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)

    class Emb(nn.Module):
        def __init__(self):
            super().__init__()
            h = cfg.hidden_size
            self.word_embeddings = nn.Embedding(cfg.vocab_size, h, padding_idx=0)
            self.position_embeddings = nn.Embedding(cfg.max_position_embeddings, h)
            self.token_type_embeddings = nn.Embedding(cfg.type_vocab_size, h)
            self.LayerNorm = nn.LayerNorm(h, eps=cfg.layer_norm_eps)
            self.token_type_embeddings_visual = nn.Embedding(cfg.type_vocab_size, h)
            self.position_embeddings_visual = nn.Embedding(cfg.max_position_embeddings, h)
            self.projection = nn.Linear(vdim, h)

        def forward(self, ids, types_, vis, vis_types):
            pos = torch.arange(ids.shape[1]).unsqueeze(0).expand_as(ids)
            text = self.word_embeddings(ids) + self.position_embeddings(pos) + self.token_type_embeddings(types_)
            v = self.projection(vis)
            v = v + self.position_embeddings_visual(torch.zeros(v.shape[:-1], dtype=torch.long)) \
                + self.token_type_embeddings_visual(vis_types)
            return self.LayerNorm(torch.cat((text, v), dim=1))

    class Base(nn.Module):
        def __init__(self):
            super().__init__()
            self.embeddings = Emb()
            self.encoder = bo.BertEncoder(cfg)
            self.pooler = bo.BertPooler(cfg)

    class Cls(nn.Module):
        def __init__(self):
            super().__init__()
            self.bert = Base()
            self.classifier = nn.Sequential(bo.BertPredictionHeadTransform(cfg), nn.Linear(cfg.hidden_size, labels))

        def forward(self, ids, input_mask, attention_mask, types_, vis, vis_types):
            ext = (1.0 - attention_mask[:, None, None, :].float()) * -10000.0
            seq = self.bert.encoder(self.bert.embeddings(ids, types_, vis, vis_types), ext)[0]
            pooled = seq[torch.arange(seq.shape[0]), input_mask.sum(1) - 2]          # pooler_strategy == "vqa"
            return {"scores": self.classifier(pooled).view(-1, labels)}

    class Wrapper(nn.Module):
        def __init__(self):
            super().__init__()
            self.model = Cls()

        def forward(self, sl):
            n = int(sl["input_mask"].sum())
            for key in ("input_ids", "input_mask", "segment_ids"):
                sl[key] = sl[key][:, :n]
            image_mask = torch.ones(sl["image_feature_0"].shape[:-1], dtype=torch.long)
            attention_mask = torch.cat((sl["input_mask"], image_mask), dim=-1)
            return self.model(sl["input_ids"], sl["input_mask"], attention_mask, sl["segment_ids"],
                              sl["image_feature_0"], torch.zeros_like(image_mask))

    model = Wrapper().eval()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
    g = torch.Generator().manual_seed(32)
    ids = torch.randint(1, 70, (1, Tpad), generator=g)
    ids[0, T:] = 0
    mask = torch.zeros(1, Tpad, dtype=torch.long)
    mask[0, :T] = 1

    def sample():
        return {"input_ids": ids.clone(), "input_mask": mask.clone(),
                "segment_ids": torch.zeros(1, Tpad, dtype=torch.long), "image_feature_0": feats}

    feats = torch.randn(1, V, vdim, generator=g)
    return model, sample, cfg, ids, mask, feats, vdim, labels


def gen_visualbert_model():
    model, sample, cfg, ids, mask, feats, vdim, labels = _build_visualbert_ref()
    arrays = dict(dims=np.array([cfg.hidden_size, cfg.num_attention_heads, cfg.intermediate_size,
                                 cfg.num_hidden_layers, cfg.vocab_size, cfg.max_position_embeddings, vdim, labels]),
                  input_ids=ids, input_mask=mask, image_feature_0=feats, scores=model(sample())["scores"],
                  out=vb_eg.SelfAttentionGenerator(model).generate_ours(sample()),
                  rollout_out=vb_eg.SelfAttentionGenerator(model).generate_rollout(sample()),
                  raw_attn_out=vb_eg.SelfAttentionGenerator(model).generate_raw_attn(sample()),
                  gradcam_out=vb_eg.SelfAttentionGenerator(model).generate_attn_gradcam(sample()))
    for k, v in model.state_dict().items():
        if not k.endswith("position_ids"):
            arrays["w__" + k] = v
    save("visualbert_model", **arrays)


# ----------------------------------------------------------------------------- perturbation evaluators (round 2)
_tensor_to = torch.Tensor.to


def _to_without_cuda(self, *args, **kwargs):
    """``x.to("cuda")`` -> ``x`` on this CPU-only box (the evaluators hard-code it, like ``.cuda()`` above)."""
    if args and (args[0] == "cuda" or (isinstance(args[0], torch.device) and args[0].type == "cuda")):
        return self
    return _tensor_to(self, *args, **kwargs)


def reference_function(rel_path, class_name, func_name, namespace):
    """Source of ``class_name.func_name`` cut out of a reference file that cannot be imported as a module (its top-level
    imports need hub downloads / mmf) and exec'd unchanged -- the same technique as the notebook cells."""
    import ast
    import textwrap
    src = open(os.path.join(REF, rel_path)).read()
    for node in ast.parse(src).body:
        if isinstance(node, ast.ClassDef) and node.name == class_name:
            for item in node.body:
                if isinstance(item, ast.FunctionDef) and item.name == func_name:
                    exec(textwrap.dedent(ast.get_source_segment(src, item)), namespace)
                    return namespace[func_name]
    raise KeyError((class_name, func_name))


def gen_detr_eval_merge():
    """``merge`` of ``DETR/datasets/coco_eval.py:170-189`` -- the module cannot be imported (pycocotools), so the function's source is cut
    out of the file and exec'd unchanged -- on the per-rank pieces of a two-rank run: image-id lists of different lengths with the
    repeated images a distributed sampler pads the last shard with, and ``evalImgs``-shaped object arrays ``[categories, areas, images]``.
    ``all_gather`` (``DETR/util/misc.py:88-128``) is a stand-in that returns the two ranks' pieces in rank order, which is its contract."""
    import ast
    import textwrap
    rel = "DETR/datasets/coco_eval.py"
    src = open(os.path.join(REF, rel)).read()
    node = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "merge")
    rng = np.random.RandomState(77)
    ids = [[17, 3, 42, 8, 23], [5, 42, 11, 3]]                       # ranks 0 / 1; 42 and 3 appear on both (sampler padding)
    evals = [rng.rand(4, 3, len(i)).astype(np.float64) for i in ids]
    pieces = {"ids": ids, "evals": evals}

    def all_gather(data):
        return pieces["ids"] if isinstance(data, list) else pieces["evals"]
    ns = {"np": np, "all_gather": all_gather}
    exec(textwrap.dedent(ast.get_source_segment(src, node)), ns)
    merged_ids, merged_evals = ns["merge"](ids[0], evals[0])
    save("detr_eval_merge", ids_rank0=np.array(ids[0]), ids_rank1=np.array(ids[1]), evals_rank0=evals[0], evals_rank1=evals[1],
         merged_ids=np.asarray(merged_ids), merged_evals=np.asarray(merged_evals))


def gen_lxmert_perturbation():
    """``ModelPert.perturbation_image`` / ``perturbation_text`` (lxmert/lxmert/perturbation.py:85-194) -- the reference's
    own method bodies, exec'd from the file -- driving the REAL reference LXMERT body of ``gen_lxmert_model`` (same
    weights: lxmert_model.npz).  The Faster R-CNN / tokenizer / dataset objects those methods touch are stand-ins that
    return the fixture's tensors; every answer-score vector of the 9 steps is recorded."""
    torch.Tensor.to = _to_without_cuda
    model, weights, inputs, cfg, T, I = _build_lxmert_ref()
    usage = types.SimpleNamespace(model=model, text_len=T, image_boxes_len=I, forward=lambda item: model(**inputs))
    R_t_t, R_t_i = lx_eg.GeneratorOurs(usage).generate_ours(None, use_lrp=False)
    cam_image, cam_text = R_t_i[0], R_t_t[0]                       # perturbation.py:242-245
    cam_image = (cam_image - cam_image.min()) / (cam_image.max() - cam_image.min())
    cam_text = (cam_text - cam_text.min()) / (cam_text.max() - cam_text.min())
    g = torch.Generator().manual_seed(23)
    label_scores = torch.rand(cfg.num_qa_labels, generator=g).round(decimals=1)          # soft VQA accuracies per answer id
    recorded = []

    def lxmert_vqa(**kw):
        kw.pop("return_dict"), kw.pop("output_attentions")
        out = model(**kw)
        recorded.append(out.question_answering_score.detach().clone())
        return out

    arrays = {"cam_image": cam_image, "cam_text": cam_text, "label_scores": label_scores}
    for name in ("perturbation_image", "perturbation_text"):
        fn = reference_function("lxmert/lxmert/perturbation.py", "ModelPert", name, {"torch": torch, "np": np})
        for positive in (False, True):
            recorded.clear()
            me = types.SimpleNamespace(
                COCO_VAL_PATH="", image_preprocess=lambda path: (None, None, None),
                frcnn=lambda *a, **k: {"normalized_boxes": inputs["visual_pos"], "roi_features": inputs["visual_feats"]},
                frcnn_cfg=types.SimpleNamespace(max_detections=I),
                lxmert_tokenizer=lambda *a, **k: types.SimpleNamespace(
                    input_ids=inputs["input_ids"], attention_mask=inputs["attention_mask"],
                    token_type_ids=inputs["token_type_ids"]),
                pert_steps=[0, 0.25, 0.5, 0.75, 0.8, 0.85, 0.9, 0.95, 1], pert_acc=[0] * 9, image_boxes_len=I,
                lxmert_vqa=lxmert_vqa, vqa_answers=[str(i) for i in range(cfg.num_qa_labels)])
            item = {"img_id": "x", "sent": "x",
                    "label": {str(i): float(v) for i, v in enumerate(label_scores) if v > 0}}
            acc = fn(me, item, cam_image.clone(), cam_text.clone(), positive)
            tag = "%s_%s" % (name.split("_")[1], "pos" if positive else "neg")
            arrays["scores_" + tag] = torch.cat(recorded)                       # [9, num_answers]
            arrays["acc_" + tag] = np.asarray(acc, dtype=np.float32)            # per-step accuracies of this ONE item
    save("lxmert_perturbation", **arrays)


def gen_visualbert_perturbation():
    """``TrainerEvaluationLoopMixinPert.evaluation_loop`` (VisualBERT/mmf/trainers/core/evaluation_loop.py:72-169) -- the
    reference's own loop, exec'd from the file -- around the reference ``SelfAttentionGenerator`` and the REAL BERT_ours
    stack of ``gen_visualbert_model`` (same weights: visualbert_model.npz), two items, all four (modality, sign)
    combinations.  Records every forward's scores and what the loop prints (incl. its ``i > num_samples`` /
    divide-by-``num_samples`` arithmetic)."""
    torch.Tensor.to = _to_without_cuda
    model, sample, cfg, ids, mask, feats, vdim, labels = _build_visualbert_ref()
    g = torch.Generator().manual_seed(33)
    targets = torch.rand(1, labels, generator=g).round(decimals=1)
    feats2 = torch.randn(2, *feats.shape[1:], generator=g)            # region features of items 1 and 2 (item 0: the fixture's)
    V = feats.shape[1]

    def batch(which):
        b = sample()
        if which:
            b["image_feature_0"] = feats2[which - 1:which].clone()
        b["image_info_0"] = {"bbox": [np.arange(V * 4, dtype=np.float32).reshape(V, 4)],
                             "max_features": torch.tensor(V).view(1), "num_boxes": [V]}
        b["tokens"] = [["t%d" % i for i in range(ids.shape[1])]]
        b["targets"] = targets
        return b

    arrays = {"targets": targets, "image_features_1_2": feats2}
    for modality in ("image", "text"):
        for positive in (False, True):
            recorded, printed = [], []

            def _forward(b):
                out = model(b)
                recorded.append(out["scores"].detach().clone())
                return {"targets": b["targets"], "scores": out["scores"]}

            pert_args = types.SimpleNamespace(args=types.SimpleNamespace(
                method="ours_no_lrp", is_positive_pert=positive, is_text_pert=(modality == "text"), num_samples=2))
            ns = {"torch": torch, "np": np, "ExplanationGenerator": vb_eg, "perturbation_arguments": pert_args,
                  "tqdm": types.SimpleNamespace(tqdm=lambda it, **k: it), "is_master": lambda: True,
                  "print": lambda *a: printed.append(a)}
            loop = reference_function("VisualBERT/mmf/trainers/core/evaluation_loop.py", "TrainerEvaluationLoopMixinPert",
                                      "evaluation_loop", ns)
            me = types.SimpleNamespace(model=model, _forward=_forward)
            # THREE items with num_samples = 2: the loop's ``i > num_samples`` test lets a third item through and the
            # printed accuracies are still divided by num_samples
            loop(me, [batch(0), batch(1), batch(2)], None)
            tag = "%s_%s" % (modality, "pos" if positive else "neg")
            arrays["scores_" + tag] = torch.cat(recorded).reshape(3, 9, labels)
            arrays["printed_step_acc_" + tag] = np.asarray(printed[-1][0], dtype=np.float64)
    save("visualbert_perturbation", **arrays)


def gen_lxmert_model_lrp():
    """The reference's REAL LRP pass of LXMERT: every ``relprop`` of ``lxmert_lrp.py`` (answer head :955-958, pooler :886-892,
    encoder :855-866, x-layer :735-740, layers :601-606, attention :422-461) over the layer library
    ``lxmert/lxmert/src/layers.py``, on the body and inputs of ``lxmert_model.npz`` (same weights).  ``LxmertModel.relprop`` /
    ``LxmertForQuestionAnswering.relprop`` (:1253-1257, :1689-1692) are restated on the wrapper (flagged glue, like its forward).
    Recorded: ``generate_ours`` with its DEFAULT arguments (``use_lrp=True``), the two LRP baselines, every attention module's
    ``attn_cam`` and the relevances the pass returns."""
    model, weights, inputs, cfg, T, I = _build_lxmert_ref()

    def model_relprop(self, cam, **kw):                                 # lxmert_lrp.py:1253-1257
        cam_lang, cam_vis = cam
        cam_lang = self.pooler.relprop(cam_lang, **kw)
        return self.encoder.relprop((cam_lang, cam_vis), **kw)

    def qa_relprop(self, cam, **kw):                                    # lxmert_lrp.py:1689-1692 (vis_shape: :1677)
        cam_lang = self.answer_head.relprop(cam, **kw)
        cam_vis = torch.zeros(1, I, cfg.hidden_size)
        return self.lxmert.relprop((cam_lang, cam_vis), **kw)

    type(model.lxmert).relprop = model_relprop
    type(model).relprop = qa_relprop
    def record(prefix):
        usage = types.SimpleNamespace(model=model, text_len=T, image_boxes_len=I, forward=lambda item: model(**inputs))
        gen = lx_eg.GeneratorOurs(usage)
        R_t_t, R_t_i = gen.generate_ours(None)                          # DEFAULT arguments: use_lrp=True
        enc = model.lxmert.encoder
        cams = {}
        for i, b in enumerate(enc.layer):
            cams["l%d" % i] = b.attention.self.get_attn_cam()
        for i, b in enumerate(enc.r_layers):
            cams["r%d" % i] = b.attention.self.get_attn_cam()
        for i, b in enumerate(enc.x_layers):
            cams["x%d_lang_self" % i] = b.lang_self_att.self.get_attn_cam()
            cams["x%d_visn_self" % i] = b.visn_self_att.self.get_attn_cam()
            cams["x%d_cross" % i] = b.visual_attention.att.get_attn_cam()
            cams["x%d_cross_copy" % i] = b.visual_attention_copy.att.get_attn_cam()
        arrays = {"cam__" + k: v.detach().clone() for k, v in cams.items()}
        arrays.update(R_t_t=R_t_t, R_t_i=R_t_i, R_i_i=gen.R_i_i, R_i_t=gen.R_i_t)
        # the relevance the pass hands back for the encoder inputs
        out = model(**inputs).question_answering_score
        index = int(out.argmax())
        one_hot = torch.zeros_like(out)
        one_hot[0, index] = 1
        model.zero_grad()
        torch.sum(one_hot * out).backward(retain_graph=True)
        cam_lang, cam_vis = model.relprop(one_hot.clone(), alpha=1)
        arrays.update(cam_lang=cam_lang, cam_vis=cam_vis, index=np.int64(index))
        base = lx_eg.GeneratorBaselines(usage)
        arrays["transformer_attr_R_t_t"], arrays["transformer_attr_R_t_i"] = base.generate_transformer_attr(None)
        arrays["partial_lrp_R_t_t"], arrays["partial_lrp_R_t_i"] = base.generate_partial_lrp(None)
        return {prefix + k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in arrays.items()}

    arrays = record("")
    # The same reference pass in float64 ("f64__" entries): safe_divide by layer outputs makes the fp32 pass itself uncertain at
    # ~1e-3 of a cam's largest entry; the distance fp32 <-> fp64 OF THE REFERENCE is the yardstick the GPU tests use.
    torch.set_default_dtype(torch.float64)
    try:
        model.double()
        inputs = {k: (v.double() if v.is_floating_point() else v) for k, v in inputs.items()}
        arrays.update(record("f64__"))
    finally:
        torch.set_default_dtype(torch.float32)
    save("lxmert_model_lrp", **arrays)


def gen_visualbert_model_lrp():
    """The reference's REAL LRP pass of VisualBERT's BERT stack: ``BERT_ours.py`` (``BertEncoder.relprop`` :152-156,
    ``BertLayer`` :506-515, ``BertAttention`` :227-232, ``BertSelfAttention`` :345-395 incl. the Add rule of the attention
    mask, ``BertPredictionHeadTransform`` :533-537) over ``layers_ours.py``, on the body and sample of
    ``visualbert_model.npz`` (same weights).  The wrapper's pooling is switched to the reference's ``IndexSelect`` module
    (``vqa_pooler``, visual_bert.py:390) and ``relprop`` of the classification model / base model / wrapper
    (visual_bert.py:398-403, :150-153, :615-616) is restated on it (flagged glue).  Recorded: the two LRP methods of
    ``SelfAttentionGenerator``, every layer's ``attn_cam``, the relevance the pass returns."""
    import importlib
    model, sample, cfg, ids, mask, feats, vdim, labels = _build_visualbert_ref()
    lo = importlib.import_module("vb_backends.layers_ours")
    cls = model.model
    head_linear = lo.Linear(cfg.hidden_size, labels)                   # visual_bert.py:312-315: an LRP ``Linear``
    head_linear.load_state_dict(cls.classifier[1].state_dict())
    cls.classifier[1] = head_linear.eval()
    cls.vqa_pooler = lo.IndexSelect()

    def cls_forward(self, ids_, input_mask, attention_mask, types_, vis, vis_types):
        ext = (1.0 - attention_mask[:, None, None, :].float()) * -10000.0
        seq = self.bert.encoder(self.bert.embeddings(ids_, types_, vis, vis_types), ext)[0]
        index = input_mask.sum(1) - 2                                   # visual_bert.py:376-390
        pooled = self.vqa_pooler(seq, 1, index)
        return {"scores": self.classifier(pooled).contiguous().view(-1, labels)}

    def cls_relprop(self, cam, **kw):                                   # visual_bert.py:398-403
        for m in reversed(self.classifier._modules.values()):
            cam = m.relprop(cam, **kw)
        cam = self.vqa_pooler.relprop(cam, **kw)
        return self.bert.encoder.relprop(cam, **kw)                     # VisualBERTBase.relprop, :150-153

    type(cls).forward = cls_forward
    type(cls).relprop = cls_relprop
    type(model).relprop = lambda self, cam, **kw: self.model.relprop(cam, **kw)          # visual_bert.py:615-616
    def record(prefix, sample):
        arrays = dict(scores=model(sample())["scores"])
        gen = vb_eg.SelfAttentionGenerator(model)
        arrays["transformer_att_out"] = gen.generate_transformer_att(sample())
        blocks = cls.bert.encoder.layer
        arrays["attn_cam"] = torch.stack([b.attention.self.get_attn_cam() for b in blocks])
        arrays["attn_grad"] = torch.stack([b.attention.self.get_attn_gradients() for b in blocks])
        arrays["partial_lrp_out"] = vb_eg.SelfAttentionGenerator(model).generate_partial_lrp(sample())
        out = model(sample())["scores"]
        index = int(out.argmax())
        one_hot = torch.zeros_like(out)
        one_hot[0, index] = 1
        arrays["cam_input"] = model.relprop(one_hot.clone(), alpha=1)
        arrays["index"] = np.int64(index)
        return {prefix + k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in arrays.items()}

    arrays = record("", sample)
    torch.set_default_dtype(torch.float64)                             # the reference's own fp32 uncertainty: see gen_lxmert_model_lrp
    try:
        model.double()

        def sample64():
            b = sample()
            b["image_feature_0"] = b["image_feature_0"].double()
            return b
        arrays.update(record("f64__", sample64))
    finally:
        torch.set_default_dtype(torch.float32)
    save("visualbert_model_lrp", **arrays)


ROUND1 = ("rules", "detr_chain", "lxmert_chain", "vit_chain", "visualbert_chain", "clip_tiny", "detr_mha",
          "detr_transformer", "lxmert_model", "visualbert_model")


def main(which):
    """``python make_golden.py`` regenerates everything; ``python make_golden.py NAME...`` only the named groups."""
    todo = {
        "rules": gen_rules,
        "detr_chain": lambda: (
            gen_detr_chain("detr_chain", 200, H=4, Ni=35, Nq=10, Le=3, Ld=3, targets=[2, 7]),
            gen_detr_chain("detr_chain_nonorm", 201, H=2, Ni=20, Nq=6, Le=2, Ld=2, targets=[0],
                           normalize_self_attention=False),
            gen_detr_chain("detr_chain_noself", 202, H=2, Ni=20, Nq=6, Le=2, Ld=2, targets=[1],
                           apply_self_in_rule_10=False)),
        "lxmert_chain": lambda: (
            gen_lxmert_chain("lxmert_chain", 250, H=4, T=7, I=12, Ll=3, Lr=2, Lx=3),
            gen_lxmert_chain("lxmert_chain_full", 251, H=12, T=14, I=36, Ll=9, Lr=5, Lx=5),
            gen_lxmert_chain("lxmert_chain_nonorm", 252, H=2, T=5, I=6, Ll=2, Lr=1, Lx=2,
                             normalize_self_attention=False)),
        "vit_chain": gen_vit_chain, "visualbert_chain": gen_visualbert_chain, "clip_tiny": gen_clip_tiny,
        "detr_mha": gen_detr_mha, "detr_transformer": gen_detr_transformer, "lxmert_model": gen_lxmert_model,
        "visualbert_model": gen_visualbert_model,
        # round 2
        "detr_chain_lrp": gen_detr_chain_lrp, "lxmert_chain_lrp": gen_lxmert_chain_lrp,
        "visualbert_chain_lrp": gen_visualbert_chain_lrp,
        "lxmert_perturbation": gen_lxmert_perturbation, "visualbert_perturbation": gen_visualbert_perturbation,
        # round 3
        "detr_transformer_lrp": gen_detr_transformer_lrp, "lrp_layers": gen_lrp_layers,
        "lxmert_model_lrp": gen_lxmert_model_lrp, "visualbert_model_lrp": gen_visualbert_model_lrp,
        # round 4
        "clip_tiny_fp16": gen_clip_tiny_fp16, "clip_vitb32_fp16": gen_clip_vitb32_fp16,
        # round 6
        "detr_eval_merge": gen_detr_eval_merge,
    }
    for name in (which or list(todo)):
        todo[name]()


if __name__ == "__main__":
    main(sys.argv[1:])
