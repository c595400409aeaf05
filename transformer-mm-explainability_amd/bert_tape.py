"""Tape forward / hand-written backward for BERT-style blocks -- what the LXMERT and VisualBERT bodies run inside an
explainability pass (``forward_tape`` / ``backward_tape`` of ``lxmert_model`` / ``visualbert_model``).

The explainability pass needs d(answer score)/d(attention probabilities) of every attention block and nothing else: no
weight gradient, no input gradient below the first explained block.  Through autograd a batched LXMERT pass is ~1100
launches of ~9 us (PyTorch builds a graph per call and runs one kernel per bias add / residual add / reshape / mask
op); here the forward keeps a small tape and the backward is the same chain of vector-Jacobian products written out, like
``clip_model.Transformer.forward_tape`` / ``backward_tape``:

  * q / k / v of a self-attention are ONE GEMM against the concatenated ``[3E, E]`` weight (cached; the attention kernels
    take the packed ``[B, N, 3, H, D]`` result in place through strides), k / v of a cross-attention one ``[2E, E]`` GEMM;
  * ``dense + residual + LayerNorm`` (``BertSelfOutput`` / ``BertOutput``: lxmert_lrp.py:464-477, 560-573;
    BERT_ours.py:345-420, 444-473) is a library GEMM (bias fused) + the fused add + LayerNorm kernel; its backward is the fused
    LayerNorm-backward kernel + one input-gradient GEMM, and every residual add of the backward is folded into a GEMM
    (``addmm``);
  * the attention core is the HIP capture op (forward writes P into the module's slab, backward writes dL/dP), reference
    cores lxmert_lrp.py:385-420, BERT_ours.py:323-343.

Everything stays fp32; GEMMs are plain library GEMMs.  Results equal the autograd route to fp32 rounding (tests pin both on
the reference's outputs).
"""
from __future__ import annotations

import math
import weakref

import torch
import torch.nn.functional as F

from . import _lib, ops

_PACKED = {}      # ids of the packed Linear modules -> (versions, W [sum_out, in], b [sum_out])


def packed_linear(mods):
    """Concatenated weight / bias of several ``nn.Linear`` with the same input (cached until a parameter changes in place)."""
    key = tuple(id(m.weight) for m in mods)
    ver = tuple(p._version for m in mods for p in (m.weight, m.bias)) + (mods[0].weight.device,)
    hit = _PACKED.get(key)
    if hit is None or hit[0] != ver:
        W = torch.cat([m.weight.detach() for m in mods], 0).contiguous()
        b = torch.cat([m.bias.detach() for m in mods], 0).contiguous()
        if hit is None:
            for m in mods:
                weakref.finalize(m.weight, _PACKED.pop, key, None)
        hit = _PACKED[key] = (ver, W, b)
    return hit[1], hit[2]


def _mask3(mask, B, Nk):
    """HF extended additive mask ``[B, 1, 1, Nk]`` -> ``[B, 1, Nk]`` fp32 (what the capture op takes), or ``None``."""
    return None if mask is None else mask.reshape(B, 1, Nk).float()


# ------------------------------------------------------------------------------------------------ attention
def attention_fwd(att, hidden, ctx=None, mask=None):
    """``BertStyleAttention`` forward on the tape.  ``ctx=None``: self-attention.  Returns ``(context [B, Nq, E], tape)``;
    the probabilities are in ``att``'s slab (``get_attn()``), the gradient slab is what ``attention_bwd`` fills."""
    B, Nq, E = hidden.shape
    H, D = att.num_attention_heads, att.attention_head_size
    if ctx is None:
        W, b = packed_linear((att.query, att.key, att.value))
        qkv = torch.addmm(b, hidden.reshape(B * Nq, E), W.t()).view(B, Nq, 3, H, D)
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
        Nk = Nq
    else:
        Nk = ctx.shape[1]
        q = torch.addmm(att.query.bias, hidden.reshape(B * Nq, E), att.query.weight.t()).view(B, Nq, H, D)
        W, b = packed_linear((att.key, att.value))
        kv = torch.addmm(b, ctx.reshape(B * Nk, ctx.shape[-1]), W.t()).view(B, Nk, 2, H, D)
        k, v = kv[:, :, 0], kv[:, :, 1]
    probs, grads = att._slabs(B, H, Nq, Nk, hidden.device)
    att._lrp_tape = None     # the tape path overwrites the slab an older LRP tape's ``probs`` points to: relprop must not mix them
    o = ops.attn_capture_fwd(q, k, v, probs, math.sqrt(D), _lib.SCALE_SCORES, _mask3(mask, B, Nk), layout="bnhd")
    att.save_attn(probs)
    att.save_attn_gradients(grads)
    return o.reshape(B, Nq, H * D), (q, k, v, o, probs, grads, ctx is None)


def attention_bwd(att, tape, d_context, need_input=True, d_hidden_res=None, d_ctx_res=None):
    """``d_context [B, Nq, E]`` -> ``(d_hidden, d_ctx)`` (``d_ctx`` is ``None`` for a self-attention: both lead to ``hidden``);
    always writes dL/dP into the module's gradient slab.  ``d_hidden_res`` / ``d_ctx_res``: contiguous gradients the results
    are ACCUMULATED INTO (in place, beta = 1 of the GEMM: the buffers are consumed).  ``need_input=False`` (lowest explained
    block): dP only."""
    q, k, v, o, probs, grads, is_self = tape
    B, Nq, H, D = q.shape
    E = H * D
    d_o = d_context.reshape(B, Nq, H, D)
    if not need_input:
        ops.attn_capture_bwd(q, k, v, probs, d_o, grads, math.sqrt(D), _lib.SCALE_SCORES, need_dqkv=False, layout="bnhd", o=o)
        return None, None
    if is_self:
        W, _ = packed_linear((att.query, att.key, att.value))
        dqkv = torch.empty(B, Nq, 3, H, D, dtype=torch.float32, device=q.device)
        ops.attn_capture_bwd(q, k, v, probs, d_o, grads, math.sqrt(D), _lib.SCALE_SCORES, need_dqkv=True, layout="bnhd",
                             out=(dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2]), o=o)
        flat = dqkv.view(B * Nq, 3 * E)
        # (in place: ``addmm(res, a, b)`` would first copy ``res`` into a new result buffer -- one device copy per call)
        d_hidden = torch.mm(flat, W) if d_hidden_res is None else d_hidden_res.view(B * Nq, -1).addmm_(flat, W)
        return d_hidden.view(B, Nq, -1), None
    Nk = k.shape[1]
    W, _ = packed_linear((att.key, att.value))
    dq = torch.empty(B, Nq, H, D, dtype=torch.float32, device=q.device)
    dkv = torch.empty(B, Nk, 2, H, D, dtype=torch.float32, device=q.device)
    ops.attn_capture_bwd(q, k, v, probs, d_o, grads, math.sqrt(D), _lib.SCALE_SCORES, need_dqkv=True, layout="bnhd",
                         out=(dq, dkv[:, :, 0], dkv[:, :, 1]), o=o)
    fq, fkv = dq.view(B * Nq, E), dkv.view(B * Nk, 2 * E)
    d_hidden = torch.mm(fq, att.query.weight) if d_hidden_res is None else \
        d_hidden_res.view(B * Nq, -1).addmm_(fq, att.query.weight)
    d_ctx = torch.mm(fkv, W) if d_ctx_res is None else d_ctx_res.view(B * Nk, -1).addmm_(fkv, W)
    return d_hidden.view(B, Nq, -1), d_ctx.view(B, Nk, -1)


# ------------------------------------------------------------------------------------------------ dense + residual + LayerNorm
def dense_add_norm_fwd(mod, hidden, residual):
    """``LayerNorm(dense(hidden) + residual)`` (a module with ``.dense`` and ``.LayerNorm``) -> ``(y, tape)``."""
    B, N, _ = hidden.shape
    d = torch.addmm(mod.dense.bias, hidden.reshape(B * N, -1), mod.dense.weight.t()).view(B, N, -1)
    ln = mod.LayerNorm
    s, y, mean, rstd = ops.add_layernorm(residual, d, ln.weight, ln.bias, ln.eps)
    return y, (s, mean, rstd)


def dense_add_norm_bwd(mod, tape, dy):
    """``dy`` -> ``(d_hidden, d_residual)``: ``d_residual`` is the gradient w.r.t. the LayerNorm input (the caller folds it
    into whatever produces the residual's gradient)."""
    s, mean, rstd = tape
    d_s = ops.layernorm_bwd_add(dy, s, mean, rstd, mod.LayerNorm.weight)
    B, N, E = d_s.shape
    return torch.mm(d_s.view(B * N, E), mod.dense.weight).view(B, N, -1), d_s


# ------------------------------------------------------------------------------------------------ feed-forward
def _act_fwd(fn, m):
    return fn(m)


def _act_bwd(fn, m, d_a):
    if fn is F.gelu:
        return torch.ops.aten.gelu_backward(d_a, m)
    if fn is F.relu:
        return d_a * (m > 0)
    if fn is torch.tanh:
        t = torch.tanh(m)
        return d_a * (1 - t * t)
    raise NotImplementedError("tape backward of activation %r" % (fn,))


def ffn_fwd(inter, output, x):
    """``output(inter(x), x)``: ``LayerNorm(dense2(act(dense1(x))) + x)`` -> ``(y, tape)``."""
    B, N, E = x.shape
    m = torch.addmm(inter.dense.bias, x.reshape(B * N, E), inter.dense.weight.t()).view(B, N, -1)
    y, t = dense_add_norm_fwd(output, _act_fwd(inter.intermediate_act_fn, m), x)
    return y, (m, t)


def ffn_bwd(inter, output, tape, dy):
    """``dy`` -> gradient w.r.t. ``x`` (both paths: through the feed-forward and the residual, one ``addmm``)."""
    m, t = tape
    d_a, d_s = dense_add_norm_bwd(output, t, dy)
    d_m = _act_bwd(inter.intermediate_act_fn, m, d_a)
    B, N, E = d_s.shape
    return d_s.view(B * N, E).addmm_(d_m.view(B * N, -1), inter.dense.weight).view(B, N, E)     # in place: d_s is ours


# ------------------------------------------------------------------------------------------------ a whole BERT layer
def self_block_fwd(att_layer, x, mask):
    """``attention`` sub-block (``.self`` = BertStyleAttention, ``.output`` = dense + residual + LayerNorm) -> ``(y, tape)``."""
    ctx, ta = attention_fwd(att_layer.self, x, None, mask)
    y, to = dense_add_norm_fwd(att_layer.output, ctx, x)
    return y, (ta, to)


def self_block_bwd(att_layer, tape, dy, need_input=True):
    ta, to = tape
    d_ctx, d_s = dense_add_norm_bwd(att_layer.output, to, dy)
    d_x, _ = attention_bwd(att_layer.self, ta, d_ctx, need_input, d_hidden_res=d_s)
    return d_x


def layer_fwd(layer, x, mask):
    """``BertLayer`` / ``LxmertLayer``: ``.attention`` (self block), ``.intermediate``, ``.output`` -> ``(y, tape)``."""
    y1, t1 = self_block_fwd(layer.attention, x, mask)
    y2, t2 = ffn_fwd(layer.intermediate, layer.output, y1)
    return y2, (t1, t2)


def layer_bwd(layer, tape, dy, need_input=True):
    t1, t2 = tape
    d_y1 = ffn_bwd(layer.intermediate, layer.output, t2, dy)
    return self_block_bwd(layer.attention, t1, d_y1, need_input)
