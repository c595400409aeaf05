"""CPU oracle (plain torch) of the LRP relevance through an attention core -- the referee of ``mmx_attn_relprop[_phase]``.

TEST INFRASTRUCTURE ONLY (see ``oracle/relevancy_np.py`` header for who may import this): the product path never imports
it; there the HIP kernels of ``csrc/attention_lrp.hip`` run (``ops.attn_relprop``), and a missing extension fails loudly.

Restates, with citations (all relative to /root/reference), what the reference computes with autograd-in-autograd:
  * ``safe_divide``  -- ``DETR/modules/layers.py:11-14`` == ``lxmert/lxmert/src/layers.py:10-13``;
  * the two ``RelPropSimple`` relprops of the attention products (``DETR/modules/layers.py:54-66`` ``einsum``;
    ``lxmert/lxmert/src/layers.py:48-61`` ``MatMul``), each result halved, in the order the modules apply them:
    ``MultiheadAttention.relprop`` (``DETR/modules/layers.py:770-781``), ``LxmertAttention.relprop``
    (``lxmert/lxmert/src/lxmert_lrp.py:422-446``), ``BertSelfAttention.relprop``
    (``VisualBERT/mmf/models/transformers/backends/BERT_ours.py:345-375``):
        S     = safe_divide(cam_O, O)                 O = P . V
        cam_P = P * (S . V^T) / 2        (what the module stores with ``save_attn_cam``)
        cam_V = V * (P^T . S) / 2
        S1    = safe_divide(cam_S, Z)                 Z = q' . k^T  (q' = scale * q for DETR, q for the BERT-style modules),
                                                      cam_S = cam_P, or the relevance the caller derived from it (BERT: after
                                                      the Add rule of the attention mask)
        cam_Q = q' * (S1 . k) / 2,       cam_K = k * (S1^T . q') / 2
Pinned on the reference's own outputs by ``tests/test_lrp_host.py`` (``tests/golden/lrp_layers.npz``) and
``tests/test_bert_lrp_host.py`` (``lxmert_model_lrp.npz`` / ``visualbert_model_lrp.npz``: the reference's real pass).
"""
from __future__ import annotations

import torch

VALUES, SCORES = 1, 2          # the phases of mmx_attn_relprop_phase (include/mmx_relevancy.h)


def safe_divide(a, b):
    den = b.clamp(min=1e-9) + b.clamp(max=1e-9)
    den = den + den.eq(0).to(den.dtype) * 1e-9
    return a / den * b.ne(0).to(b.dtype)


def _bh(t):                     # [B, N, H, D] -> [B, H, N, D]
    return t.permute(0, 2, 1, 3)


def _back(t):
    return t.permute(0, 2, 1, 3).contiguous()


def core(tape, cam_o, cam_scores=None, phase=VALUES | SCORES, scale=1.0):
    """``tape``: ``q, k, v, o [B, N, H, D]``, ``probs [B, H, Nq, Nk]``; ``cam_o [B, Nq, H, D]``.  Returns ``(cam_probs, cam_q,
    cam_k, cam_v)`` (``None`` for a phase not run); q / k / v cams in ``[B, N, H, D]``."""
    q, k, v, o, probs = _bh(tape["q"]) * scale, _bh(tape["k"]), _bh(tape["v"]), _bh(tape["o"]), tape["probs"]
    cam_p = cam_q = cam_k = cam_v = None
    if phase & VALUES:
        S = safe_divide(_bh(cam_o), o)
        cam_p = probs * torch.matmul(S, v.transpose(-1, -2)) / 2
        cam_v = _back(v * torch.matmul(probs.transpose(-1, -2), S) / 2)
    if phase & SCORES:
        S1 = safe_divide(cam_p if cam_scores is None else cam_scores, torch.matmul(q, k.transpose(-1, -2)))
        cam_q = _back(q * torch.matmul(S1, k) / 2)
        cam_k = _back(k * torch.matmul(S1.transpose(-1, -2), q) / 2)
    return cam_p, cam_q, cam_k, cam_v


def detr_core(tape):
    """The callable ``lrp.mha_relprop`` takes (DETR: ``q * scaling`` first, fused phases): ``cam_o -> (cam_probs, cam_q, cam_k, cam_v)``."""
    return lambda cam_o: core(tape, cam_o, scale=tape["scale"])
