"""LRP pass (``model.relprop(one_hot, alpha=1)``) of the BERT-style bodies -- LXMERT and VisualBERT -- SURVEY.md section 8 row f4.

The reference builds both bodies on an LRP layer library (``lxmert/lxmert/src/layers.py`` ==
``VisualBERT/mmf/models/transformers/backends/layers_ours.py``): every layer keeps its input in a forward hook and every
``relprop`` re-runs the layer and calls ``torch.autograd.grad``.  What the generators read from that pass is one tensor per
attention module: ``get_attn_cam()``, the relevance of the attention probabilities (``lxmert_lrp.py:432``,
``BERT_ours.py:358``).  Here the modules of ``lxmert_model`` / ``visualbert_model`` keep a small tape of detached
activations while grad mode is on (``_lrp_tape``), and the pass is the same chain of rules in closed form (``lrp.py``) around
the HIP attention-core kernels (``ops.attn_relprop`` -> ``csrc/attention_lrp.hip``):

  ``attention_relprop``        ``LxmertAttention.relprop`` (lxmert_lrp.py:422-461) / ``BertSelfAttention.relprop``
                               (BERT_ours.py:345-395, ``mask_rule=True``: the Add rule of [scores, attention_mask] between
                               the two matmul relprops; LXMERT's module never assigns its ``attention_mask`` slot, so its pass
                               skips that rule)
  ``dense_add_norm_relprop``   ``LxmertAttentionOutput`` / ``LxmertOutput`` / ``BertSelfOutput`` / ``BertOutput`` ``.relprop``
                               (lxmert_lrp.py:479-486, 575-581; BERT_ours.py:412-420, 459-473)
  ``ffn_relprop``              intermediate + output + the Clone in front (lxmert_lrp.py:554-557, 601-606; BERT_ours.py:436-441,
                               506-515)
  ``self_layer_relprop``       ``LxmertSelfAttentionLayer.relprop`` (lxmert_lrp.py:535-540)
  ``cross_layer_relprop``      ``LxmertCrossAttentionLayer.relprop`` (lxmert_lrp.py:505-510)
  ``bert_attention_relprop``   ``BertAttention.relprop`` (BERT_ours.py:227-232)
  ``pooler_relprop``           ``LxmertPooler.relprop`` (lxmert_lrp.py:886-892)

The rule variants differ from DETR's library in two places: ``Linear.relprop`` has no renormalisation
(lxmert/src/layers.py:230-252 vs DETR/modules/layers.py:432) and the matmuls are ``MatMul`` modules (same ``RelPropSimple`` rule as
``einsum``).  LayerNorm, GELU / Tanh, Softmax and Dropout pass relevance through (``RelProp.relprop``, layers.py:45-46).

The reference runs one sample per pass; every whole-tensor sum of its rules (``Add.relprop``) is taken PER SAMPLE here, so a
batch gives what the reference gives item by item.  ``core``: the attention-core implementation -- the HIP op (``core_hip``)
unless a caller hands in another one (the CPU test suite passes the plain-torch referee of ``oracle/lrp_torch.py``).
"""
from __future__ import annotations

import math

import torch

from . import _lib, lrp, ops

VALUES, SCORES = _lib.LRP_VALUES, _lib.LRP_SCORES



def tape_of(module):
    """The detached activations a forward in grad mode left on ``module``; a clear error instead of a stale / missing tape
    (every forward -- eager, ``no_grad`` or the hand-written tape path -- resets it)."""
    t = getattr(module, "_lrp_tape", None)
    if t is None:
        raise RuntimeError("%s has no LRP tape: relprop needs a forward run through the module path with gradients enabled "
                           "right before it (a no_grad forward or the tape path forward_tape() clears the tapes)"
                           % type(module).__name__)
    return t


def _lin(R, X, weight):
    return lrp.linear_relprop(R, X, weight, normalize=False)


def core_hip(t, cam_o, cam_scores=None, phase=VALUES | SCORES):
    """``(cam_probs, cam_q, cam_k, cam_v)`` of one attention module's tape on the HIP kernels (``None`` for a phase not run)."""
    return ops.attn_relprop(t["q"], t["k"], t["v"], t["probs"], t["o"], cam_o, 1.0, _lib.SCALE_SCORES, "bnhd", phase, cam_scores)


def _mask_add_relprop(cam_p, t):
    """``Add.relprop`` of ``[attention_scores / sqrt(d), attention_mask]`` (BERT_ours.py:325-328, 366-368; layers_ours.py:
    Add): the relevance that stays on the scores.  The mask is ``[B, 1, 1, Nk]``: autograd sums its share over heads and
    query rows."""
    D = t["q"].shape[-1]
    a = torch.matmul(t["q"].permute(0, 2, 1, 3), t["k"].permute(0, 2, 3, 1)) / math.sqrt(D)
    b = t["mask"].reshape(a.shape[0], 1, 1, a.shape[-1]).to(a.dtype)
    S = lrp.safe_divide(cam_p, a + b)
    ra, rb = a * S, b * S.sum(dim=(1, 2), keepdim=True)
    sa, sb, total = (lrp.sample_sum(x) for x in (ra, rb, cam_p))
    fa = lrp.safe_divide(sa.abs(), sa.abs() + sb.abs()) * total
    return (ra * lrp.safe_divide(fa, sa)).contiguous()


def attention_relprop(att, t, cam_ctx, core=None, mask_rule=False):
    """Relevance of an attention module's context output ``cam_ctx [B, Nq, E]`` -> ``(cam_hidden [B, Nq, E], cam_key_input,
    cam_value_input [B, Nk, E])`` (relevances of the inputs of the query / key / value projections, before the caller's Clone
    rule); stores the relevance of the probabilities with ``att.save_attn_cam``."""
    core = core_hip if core is None else core
    B, Nq, H, D = t["q"].shape
    E = H * D
    cam_o = cam_ctx.reshape(B, Nq, H, D)
    if mask_rule and t["mask"] is not None:
        cam_p, _, _, cam_v = core(t, cam_o, None, VALUES)
        _, cam_q, cam_k, _ = core(t, None, _mask_add_relprop(cam_p, t), SCORES)
    else:
        cam_p, cam_q, cam_k, cam_v = core(t, cam_o)
    att.save_attn_cam(cam_p)
    return (_lin(cam_q.reshape(B, Nq, E), t["hidden"], att.query.weight),
            _lin(cam_k.reshape(B, -1, E), t["context"], att.key.weight),
            _lin(cam_v.reshape(B, -1, E), t["context"], att.value.weight))


def dense_add_norm_relprop(mod, t, cam):
    """``LayerNorm(dense(hidden) + residual)`` -> ``(cam_hidden, cam_residual)``; ``t = (hidden, dense(hidden), residual)``."""
    hidden, d, residual = t
    cam1, cam2 = lrp.add_relprop(cam, d, residual, per_sample=True)
    return _lin(cam1, hidden, mod.dense.weight), cam2


def ffn_relprop(inter, output, t_inter, t_out, cam):
    """``output(inter(x), x)`` -> relevance of ``x`` (the layer's Clone rule joins the two paths)."""
    cam1, cam2 = dense_add_norm_relprop(output, t_out, cam)
    cam1 = _lin(cam1, t_inter, inter.dense.weight)                        # the activation passes relevance through
    return lrp.clone_relprop((cam1, cam2), t_inter)


def self_layer_relprop(layer, cam, core=None):
    """``LxmertSelfAttentionLayer`` (``.self``, ``.output``) -> relevance of its input (Clone of 3: query, context, residual)."""
    t = tape_of(layer.self)
    cam_out, cam_res = dense_add_norm_relprop(layer.output, layer._lrp_out, cam)
    cam_h, cam_k, cam_v = attention_relprop(layer.self, t, cam_out, core)
    cam_ctx = lrp.clone_relprop((cam_k, cam_v), t["context"])              # LxmertAttention's own Clone (key, value)
    return lrp.clone_relprop((cam_h, cam_ctx, cam_res), t["hidden"])


def cross_layer_relprop(layer, cam, core=None):
    """``LxmertCrossAttentionLayer`` (``.att``, ``.output``) -> ``(cam_input, cam_context)``."""
    t = tape_of(layer.att)
    cam_out, cam_res = dense_add_norm_relprop(layer.output, layer._lrp_out, cam)
    cam_h, cam_k, cam_v = attention_relprop(layer.att, t, cam_out, core)
    return lrp.clone_relprop((cam_h, cam_res), t["hidden"]), lrp.clone_relprop((cam_k, cam_v), t["context"])


def lxmert_layer_relprop(layer, cam, core=None):
    """``LxmertLayer.relprop`` (lxmert_lrp.py:601-606)."""
    cam = ffn_relprop(layer.intermediate, layer.output, tape_of(layer.intermediate), tape_of(layer.output), cam)
    return self_layer_relprop(layer.attention, cam, core)


def bert_attention_relprop(att_layer, cam, core=None):
    """``BertAttention.relprop``: the module's Clone of 2 around ``BertSelfAttention``'s own Clone of 3."""
    t = tape_of(att_layer.self)
    cam_out, cam_res = dense_add_norm_relprop(att_layer.output, tape_of(att_layer.output), cam)
    cams = attention_relprop(att_layer.self, t, cam_out, core, mask_rule=True)
    return lrp.clone_relprop((lrp.clone_relprop(cams, t["hidden"]), cam_res), t["hidden"])


def bert_layer_relprop(layer, cam, core=None):
    """``BertLayer.relprop`` (BERT_ours.py:506-515)."""
    cam = ffn_relprop(layer.intermediate, layer.output, tape_of(layer.intermediate), tape_of(layer.output), cam)
    return bert_attention_relprop(layer.attention, cam, core)


def pooler_relprop(pooler, cam):
    """``LxmertPooler.relprop``: Tanh passes through, the dense rule, then ``IndexSelect`` of token 0."""
    hidden = tape_of(pooler)
    cam = _lin(cam, hidden[:, 0], pooler.dense.weight).unsqueeze(1)
    return lrp.index_select_relprop(cam, hidden, 1, torch.zeros(1, dtype=torch.long, device=hidden.device))
