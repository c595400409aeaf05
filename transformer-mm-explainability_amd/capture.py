"""Preallocated capture slabs + the autograd wrapper of the HIP attention-capture op.

Replaces the reference's Python hook pair (``save_attn`` + ``Tensor.register_hook(save_attn_gradients)``;
CLIP/clip/auxilary.py:247-250, DETR/modules/layers.py:758-759, lxmert_lrp.py:407-408, BERT_ours.py:332-333):
the forward kernel writes the softmax probabilities of layer ``l`` straight into ``probs[l]`` and the backward
kernel writes ``d loss / d probs`` straight into ``grads[l]``.  Nothing is copied, no temporaries of size
``[B*H, N, N]`` exist besides these two slabs.

Layout in HBM: ``probs`` and ``grads`` are each one contiguous ``[L, B, H, Nq, Nk]`` fp32 tensor, so a layer is
a contiguous ``[B*H, Nq, Nk]`` slab (index ``b*H + h`` -- the reference's CLIP layout) and a sample's heads are
contiguous (what the chain kernel streams).
"""
from __future__ import annotations

import torch

from . import _lib, ops


class CaptureBuffers:
    """Two ``[L, B, H, Nq, Nk]`` slabs (probabilities and their gradients) for one tower; fp32, or fp16 / bf16 for the
    long-sequence towers (half the resident bytes and half the rule kernels' traffic; the rules accumulate in fp32)."""

    def __init__(self, n_layers, batch, heads, n_q, n_k=None, device="cuda", shared_probs=False, dtype=torch.float32,
                 grads=True):
        """``shared_probs``: the probabilities come from ONE forward pass shared by the whole batch (``probs`` has batch
        1, ``grads`` has batch ``batch``) -- CLIP ``interpret`` repeats one image ``batch`` times."""
        n_k = n_q if n_k is None else n_k
        self.shape = (n_layers, batch, heads, n_q, n_k)
        self.shared_probs = shared_probs
        self.dtype = dtype
        p_shape = (n_layers, 1 if shared_probs else batch, heads, n_q, n_k)
        self.probs = torch.empty(p_shape, dtype=dtype, device=device)      # (no slack: no kernel reads outside a slab)
        # grads=False: probabilities only (row-relevancy mode of the backward never stores dP)
        self.grads = torch.empty(self.shape, dtype=dtype, device=device) if grads else None

    @property
    def n_layers(self):
        return self.shape[0]

    @property
    def batch(self):
        return self.shape[1]

    def matches(self, n_layers, batch, heads, n_q, n_k, device, shared_probs=False, dtype=torch.float32, grads=True):
        return self.shape == (n_layers, batch, heads, n_q, n_k) and self.probs.device == torch.device(device) \
            and self.shared_probs == shared_probs and self.dtype == dtype and (self.grads is not None) == bool(grads)

    def layer_probs(self, l):
        """``[B*H, Nq, Nk]`` view, the shape the reference's ``attn_probs`` / ``get_attn()`` has."""
        _, b, h, nq, nk = self.shape
        return self.probs[l].view(-1, nq, nk)      # [B*H, Nq, Nk] ([H, Nq, Nk] when the forward is shared)

    def layer_grads(self, l):
        """``[B*H, Nq, Nk]`` view of layer ``l``'s gradient slab; ``None`` for probabilities-only buffers (``grads=False``)."""
        if self.grads is None:
            return None
        _, b, h, nq, nk = self.shape
        return self.grads[l].view(b * h, nq, nk)

    def nbytes(self):
        return (self.probs.numel() + (self.grads.numel() if self.grads is not None else 0)) * self.probs.element_size()


class _AttnCapturePacked(torch.autograd.Function):
    """Self-attention on a packed ``qkv [B, N, 3, H, D]`` tensor; P and dP land in the given slabs."""

    @staticmethod
    def forward(ctx, qkv, mask, probs_slab, grads_slab, scale, scale_mode, need_dqkv, grad_hook):
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
        o = ops.attn_capture_fwd(q, k, v, probs_slab, scale, scale_mode, mask, layout="bnhd")
        ctx.save_for_backward(qkv, o)        # O: lets the long-sequence backward skip its delta sweep (rowsum(dO * O))
        ctx.probs, ctx.grads = probs_slab, grads_slab
        ctx.cfg = (scale, scale_mode, need_dqkv, grad_hook)
        return o

    @staticmethod
    def backward(ctx, d_o):
        qkv, o = ctx.saved_tensors
        scale, scale_mode, need_dqkv, grad_hook = ctx.cfg
        need = bool(need_dqkv and ctx.needs_input_grad[0])
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
        dqkv = None
        out = None
        if need:
            dqkv = torch.empty_like(qkv)
            out = (dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2])
        ops.attn_capture_bwd(q, k, v, ctx.probs, d_o, ctx.grads, scale, scale_mode, need_dqkv=need, layout="bnhd",
                             out=out, o=o)
        if grad_hook is not None:
            grad_hook(ctx.grads)
        return dqkv, None, None, None, None, None, None, None


class _AttnCapture(torch.autograd.Function):
    """General (cross-)attention: separate ``q [B, Nq, H, D]``, ``k``/``v [B, Nk, H, D]`` (strided views ok)."""

    @staticmethod
    def forward(ctx, q, k, v, mask, probs_slab, grads_slab, scale, scale_mode, grad_hook, anchor):
        o = ops.attn_capture_fwd(q, k, v, probs_slab, scale, scale_mode, mask, layout="bnhd")
        ctx.save_for_backward(q, k, v, o)
        ctx.probs, ctx.grads = probs_slab, grads_slab
        ctx.cfg = (scale, scale_mode, grad_hook)
        return o

    @staticmethod
    def backward(ctx, d_o):
        q, k, v, o = ctx.saved_tensors
        scale, scale_mode, grad_hook = ctx.cfg
        need = any(ctx.needs_input_grad[:3])
        dq, dk, dv = ops.attn_capture_bwd(q, k, v, ctx.probs, d_o, ctx.grads, scale, scale_mode, need_dqkv=need,
                                          layout="bnhd", o=o)
        if grad_hook is not None:
            grad_hook(ctx.grads)
        return dq, dk, dv, None, None, None, None, None, None, None


def attention_capture_packed(qkv, probs_slab, grads_slab, scale, mask=None, scale_mode=_lib.SCALE_Q_FIRST,
                             need_dqkv=True, grad_hook=None):
    """``qkv``: ``[B, N, 3, H, D]`` fp32 (a view of the in-projection output).  Returns ``O [B, N, H, D]``."""
    return _AttnCapturePacked.apply(qkv, mask, probs_slab, grads_slab, scale, scale_mode, need_dqkv, grad_hook)


def attention_capture(q, k, v, probs_slab, grads_slab, scale, mask=None, scale_mode=_lib.SCALE_Q_FIRST,
                      grad_hook=None):
    """``q [B, Nq, H, D]``, ``k``/``v [B, Nk, H, D]`` fp32.  Returns ``O [B, Nq, H, D]``.

    dL/dP is defined -- and read by the rules -- even when q, k and v are all constants of the graph (DETR's first
    decoder self-attention under frozen parameters: ``tgt = 0`` and ``query_pos`` is a frozen embedding; the reference's
    tensor hook fires there because its weights require grad).  Autograd would drop such a node, so it is tied to the
    graph through a scalar anchor: the backward kernel then runs whenever the loss depends on ``O``.
    """
    anchor = None
    if torch.is_grad_enabled() and not (q.requires_grad or k.requires_grad or v.requires_grad):
        anchor = torch.zeros((), requires_grad=True)          # host scalar: no launch, its gradient is never formed
    return _AttnCapture.apply(q, k, v, mask, probs_slab, grads_slab, scale, scale_mode, grad_hook, anchor)
