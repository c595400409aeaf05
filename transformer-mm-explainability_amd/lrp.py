"""Alpha-beta LRP rules of the reference's layer library (``DETR/modules/layers.py``), closed form -- SURVEY.md section 8 row f4.

The reference implements every ``relprop`` as autograd-in-autograd: a forward hook keeps the layer input ``X``, ``relprop``
re-runs the layer on clamped copies of ``X`` and calls ``torch.autograd.grad`` for the vector-Jacobian product (4 ``F.linear``
+ 4 ``autograd.grad`` per ``Linear``, ``layers.py:409-437``).  For the layer types the DETR / LXMERT / BERT transformers are
built from, those vector-Jacobian products are plain matrix products, written out here:

  ``safe_divide``            layers.py:11-14
  ``linear_relprop``         ``Linear.relprop``, layers.py:409-437 (alpha = 1: the inhibitor term is multiplied by beta = 0)
  ``add_relprop``            ``Add.relprop``, layers.py:197-222
  ``clone_relprop``          ``Clone.relprop``, layers.py:257-267
  ``index_select_relprop``   ``IndexSelect.relprop``, layers.py:232-244
  attention core             ``MultiheadAttention.relprop``, layers.py:770-801: the two ``einsum`` relprops
                             (``RelPropSimple``, layers.py:54-66) run in the HIP kernels of ``csrc/attention_lrp.hip``
                             (``ops.attn_relprop``) on the tiles the capture op uses; ``mha_relprop`` below is the module-level
                             schedule around them (out_proj / q_proj / k_proj / v_proj relprops and the q/k rescale branch).
                             (The CPU test suite passes the plain-torch referee of ``oracle/lrp_torch.py`` as ``attn_core``.)
LayerNorm, ReLU / GELU, Softmax, Dropout and ``WithPosEmbd`` pass relevance through unchanged (``RelProp.relprop``,
layers.py:46-47, 110-111).  Everything is sync-free (global sums stay device scalars).
"""
from __future__ import annotations

import torch

from . import ops


def safe_divide(a, b):
    """layers.py:11-14: ``a / (b + 1e-9)`` (the two clamps add up to that), 0 where ``b == 0``."""
    den = b.clamp(min=1e-9) + b.clamp(max=1e-9)
    den = den + den.eq(0).to(den.dtype) * 1e-9
    return a / den * b.ne(0).to(b.dtype)


def sample_sum(t):
    """Sum over everything but the batch dimension, kept broadcastable: what a whole-tensor ``.sum()`` of the reference's
    one-sample pass is for each sample of a batch."""
    return t.sum(dim=tuple(range(1, t.dim())), keepdim=True) if t.dim() > 1 else t


def linear_relprop(R, X, weight, alpha=1, normalize=True):
    """``Linear.relprop`` (layers.py:409-437).  ``X [..., in]``, ``weight [out, in]``, ``R [..., out]`` -> ``[..., in]``.
    ``normalize=False``: the LXMERT / VisualBERT flavour of the library (lxmert/lxmert/src/layers.py:230-252), the same rule
    without DETR's closing ``R_out * safe_divide(R.sum(), R_out.sum())`` (layers.py:432)."""
    if alpha != 1:
        raise NotImplementedError("the generators call relprop with alpha = 1 (DETR/modules/ExplanationGenerator.py:148)")
    if ops.lrp_fusable(R, X, weight):          # 2 GEMMs + 3-4 launches (csrc/lrp_kernels.hip) instead of 4 GEMMs + ~35
        return ops.lrp_linear(R, X, weight, normalize)
    pw, nw = weight.clamp(min=0), weight.clamp(max=0)
    px, nx = X.clamp(min=0), X.clamp(max=0)
    Z = torch.matmul(px, pw.t()) + torch.matmul(nx, nw.t())
    S = safe_divide(R, Z)
    out = px * torch.matmul(S, pw) + nx * torch.matmul(S, nw)                  # activator relevances; beta = alpha - 1 = 0
    return out * safe_divide(R.sum(), out.sum()) if normalize else out


def add_relprop(R, a, b, per_sample=False):
    """``Add.relprop`` (layers.py:197-222) -> ``(R_a, R_b)``.  ``per_sample``: the three whole-tensor sums per batch item."""
    if ops.lrp_fusable(R, a, b) and R.shape == a.shape == b.shape:
        fused = ops.lrp_add(R, a, b, per_sample)
        if fused is not None:
            return fused
    S = safe_divide(R, a + b)
    ra, rb = a * S, b * S
    total_of = sample_sum if per_sample else torch.sum
    sa, sb, total = total_of(ra), total_of(rb), total_of(R)
    fa = safe_divide(sa.abs(), sa.abs() + sb.abs()) * total
    fb = safe_divide(sb.abs(), sa.abs() + sb.abs()) * total
    return ra * safe_divide(fa, sa), rb * safe_divide(fb, sb)


def clone_relprop(Rs, X):
    """``Clone.relprop`` (layers.py:257-267): ``X * sum_i safe_divide(R_i, X)``."""
    Rs = list(Rs)
    if 1 <= len(Rs) <= 8 and ops.lrp_fusable(X, *Rs) and all(r.shape == X.shape for r in Rs):
        return ops.lrp_clone(Rs, X)
    C = None
    for r in Rs:
        s = safe_divide(r, X)
        C = s if C is None else C + s
    return X * C


def index_select_relprop(R, X, dim, indices):
    """``IndexSelect.relprop`` (layers.py:232-244): relevance lands on the selected slices of ``X``, zero elsewhere."""
    Z = torch.index_select(X, dim, indices)
    C = torch.zeros_like(X).index_add_(dim, indices, safe_divide(R, Z))
    return X * C


def _all_zero(t):
    """``t.min() == t.max() == 0`` (layers.py:791, 796) as a device boolean."""
    return (t.min() == 0) & (t.max() == 0)


def mha_relprop(cam_out, tape, weights, attn_core):
    """``MultiheadAttention.relprop`` (layers.py:770-801), batch-first.

    ``cam_out [B, T, E]``: relevance of the module output.  ``tape``: ``dict(query [B,T,E], key [B,S,E], value [B,S,E],
    q / k / v [B, N, H, D] projected (q unscaled), o [B, T, H, D], probs [B, H, T, S], scale)`` of the forward.
    ``weights``: ``(Wq, Wk, Wv, Wo)``.  ``attn_core(cam_o [B,T,H,D]) -> (cam_probs [B,H,T,S], cam_q, cam_k, cam_v [B,N,H,D])``:
    the two einsum relprops INCLUDING their ``/ 2`` (layers.py:773-781).  Returns ``(cam_query, cam_key, cam_value,
    cam_probs)``; the caller stores ``cam_probs`` with ``save_attn_cam`` (layers.py:776)."""
    Wq, Wk, Wv, Wo = weights
    B, T, H, D = tape["o"].shape
    E = H * D
    cam_o = linear_relprop(cam_out, tape["o"].reshape(B, T, E), Wo).view(B, T, H, D)
    cam_probs, cam_q, cam_k, cam_v = attn_core(cam_o)
    cam_v, cam_k, cam_q = cam_v.reshape(B, -1, E), cam_k.reshape(B, -1, E), cam_q.reshape(B, T, E)
    cam_v_pre = cam_v
    # ONE decision for both the bookkeeping (pre_zero) and the rescale branch, taken on everything the fused launches will read:
    # the projections below keep dtype and device, so the four streams are fusable after them iff they are now
    fused = ops.lrp_fusable(cam_v, cam_k, cam_q, cam_o)
    pre_zero = None if fused else _all_zero(cam_v)
    cam_v = linear_relprop(cam_v, tape["value"], Wv)
    cam_k = linear_relprop(cam_k, tape["key"], Wk)
    cam_q = linear_relprop(cam_q, tape["query"], Wq)
    # layers.py:791-799: a value stream that carries no relevance through its projection (decoder layer 0: value = 0)
    # hands the head-level relevance to the query / key streams, split by their share of the total
    if fused and ops.lrp_fusable(cam_v, cam_k, cam_q):                  # two launches instead of ~35 (csrc/lrp_kernels.hip)
        cam_k, cam_q = ops.lrp_mha_rescale(cam_v_pre, cam_v, cam_k, cam_q, cam_o)
        return cam_q, cam_k, cam_v, cam_probs
    if pre_zero is None:                                                # (a projection changed dtype / device: torch formulation)
        pre_zero = _all_zero(cam_v_pre)
    rescale = _all_zero(cam_v) & ~pre_zero
    ks, qs, total = cam_k.sum(), cam_q.sum(), cam_o.sum()
    kf = safe_divide(ks.abs(), ks.abs() + qs.abs()) * total
    qf = safe_divide(qs.abs(), ks.abs() + qs.abs()) * total
    cam_k = torch.where(rescale, cam_k * safe_divide(kf, ks), cam_k)
    cam_q = torch.where(rescale, cam_q * safe_divide(qf, qs), cam_q)
    return cam_q, cam_k, cam_v, cam_probs
