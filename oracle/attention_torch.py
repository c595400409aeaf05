"""CPU oracle (torch fp32) of the reference's HOOKED attention cores -- what ``save_attn`` / ``save_attn_gradients`` see.

TEST INFRASTRUCTURE ONLY (see ``oracle/relevancy_np.py`` header for who may import this): the product path never
imports it; the HIP capture op (``mmx_attn_capture_fwd/bwd``) is what runs there.

Restates, with citations (all relative to /root/reference):
  * ``SCALE_Q_FIRST``  -- ``q = q * scaling`` before ``bmm(q, k^T)``: CLIP/clip/auxilary.py:153,225 and
    DETR/modules/layers.py:737-746 (``q = q * scaling``, then the ``bmm`` einsum);
  * ``SCALE_SCORES``   -- ``scores / sqrt(d)`` after the product: lxmert/lxmert/src/lxmert_lrp.py:398-400 and
    VisualBERT/mmf/models/transformers/backends/BERT_ours.py:323-326;
  * additive mask, then ``softmax(dim=-1)``; the softmax output is the tensor handed to ``save_attn`` and the tensor
    whose ``register_hook`` gradient is handed to ``save_attn_gradients`` (auxilary.py:243-250, layers.py:751-755,
    lxmert_lrp.py:405-408, BERT_ours.py:330-333); ``O = P @ V``.
``detr_mha`` is the reference's whole hooked ``MultiheadAttention.forward`` (DETR/modules/layers.py:727-765: three
separate projections, heads folded into the batch, ``out_proj``).  Pinned by
``tests/test_oracle_golden.py::test_attention_oracle_vs_reference_mha`` against ``tests/golden/detr_mha.npz`` (outputs of
the reference's own module).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

SCALE_Q_FIRST = 0
SCALE_SCORES = 1


def core(q, k, v, scale, scale_mode=SCALE_Q_FIRST, mask=None):
    """``q [.., Nq, D]``, ``k, v [.., Nk, D]`` -> ``(P, O)``, both attached to the autograd graph."""
    if scale_mode == SCALE_Q_FIRST:
        s = torch.matmul(q * scale, k.transpose(-1, -2))
    else:
        s = torch.matmul(q, k.transpose(-1, -2)) / scale
    if mask is not None:
        s = s + mask
    p = F.softmax(s, dim=-1)
    return p, torch.matmul(p, v)


def capture(q, k, v, d_o, scale, scale_mode=SCALE_Q_FIRST, mask=None):
    """The capture op's contract on ``[B, H, N, D]`` operands: returns ``(P, O, dP, dq, dk, dv)`` (detached fp32)."""
    q, k, v = (t.detach().float().clone().requires_grad_(True) for t in (q, k, v))
    p, o = core(q, k, v, scale, scale_mode, mask)
    dp, dq, dk, dv = torch.autograd.grad(o, [p, q, k, v], grad_outputs=d_o.float())
    return p.detach(), o.detach(), dp, dq, dk, dv


def detr_mha(sd, query, key, value, num_heads, upstream):
    """DETR/modules/layers.py:727-765 (no masks, eval): ``query [T, B, E]``, ``key/value [S, B, E]``.
    Returns ``dict(out, attn [B*H, T, S], attn_grad, dquery, dkey, dvalue)``."""
    query, key, value = (t.detach().float().clone().requires_grad_(True) for t in (query, key, value))
    T, B, E = query.shape
    S = key.shape[0]
    d = E // num_heads
    q = F.linear(query, sd["q_proj.weight"], sd["q_proj.bias"])
    k = F.linear(key, sd["k_proj.weight"], sd["k_proj.bias"])
    v = F.linear(value, sd["v_proj.weight"], sd["v_proj.bias"])
    q = q.contiguous().view(T, B * num_heads, d).transpose(0, 1)       # layers.py:741-743
    k = k.contiguous().view(S, B * num_heads, d).transpose(0, 1)
    v = v.contiguous().view(S, B * num_heads, d).transpose(0, 1)
    p, o = core(q, k, v, float(d) ** -0.5, SCALE_Q_FIRST)
    out = F.linear(o.transpose(0, 1).contiguous().view(T, B, E), sd["out_proj.weight"], sd["out_proj.bias"])
    dp, dq, dk, dv = torch.autograd.grad(out, [p, query, key, value], grad_outputs=upstream.float())
    return dict(out=out.detach(), attn=p.detach(), attn_grad=dp, dquery=dq, dkey=dk, dvalue=dv)
