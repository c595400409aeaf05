"""DETR's transformer (encoder-decoder) and detection heads on the HIP capture op.

This is the part of the DETR model the relevancy path runs through: six encoder
layers of self-attention over the image tokens, six decoder layers of query self-attention + query->image
cross-attention, post-norm by default.  Parameter names follow ``DETR/models/transformer.py`` and
``DETR/models/detr.py:20-41`` so a DETR-R50 checkpoint's ``transformer.*``, ``class_embed.*``, ``bbox_embed.*``,
``query_embed.*`` and ``input_proj.*`` entries load unchanged.  The CNN backbone is torchvision's ResNet and is not
part of the hot path: ``DETRFromFeatures`` takes the backbone's last feature map.

Differences from the reference that a caller can observe:
  * every attention block is ``attention_modules.MultiheadAttention`` -- P and dL/dP land in device slabs written by
    the HIP kernels, ``get_attn()`` / ``get_attn_gradients()`` return views of them, there are no Python hooks;
  * padding masks are accepted and ignored, exactly like the reference's hooked MHA (``DETR/modules/layers.py:728-756``);
  * eval mode only (dropout is the identity); ``relprop`` (the LRP pass of ``DETR/models/detr.py:79-92`` / ``transformer.py``) is
    closed-form matrix products + HIP attention-core kernels instead of autograd-in-autograd (``lrp.py``), post-norm layers;
  * the reference hard-codes decoder layer index 5 for ``pred_logits`` (``detr.py:64``); here it is the last layer.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import lrp, ops
from .attention_modules import MultiheadAttention


def _detached(*tensors):
    """Activations an LRP pass (``relprop``) reads, kept as detached references when the forward runs in grad mode (the
    explainability pass); ``None`` under ``torch.no_grad()`` so that plain inference pins nothing."""
    return tuple(t.detach() for t in tensors) if torch.is_grad_enabled() else None


def _lrp_tape(module):
    tape = getattr(module, "_lrp", None)
    if tape is None:
        raise RuntimeError("%s.relprop needs the activations of a forward pass run with gradients enabled (the generators' "
                           "explainability pass); run model(img) first" % type(module).__name__)
    return tape


def _with_pos(x, pos):
    return x if pos is None else x + pos


def _activation(name):
    try:
        return {"relu": F.relu, "gelu": F.gelu, "glu": F.glu}[name]
    except KeyError:
        raise RuntimeError(f"activation should be relu/gelu/glu, not {name}") from None


def _ln_fwd(ln, x):
    """LayerNorm forward that keeps what the hand-written backward needs: ``(y, (x, mean, rstd))``."""
    y, mean, rstd = torch.native_layer_norm(x, (x.shape[-1],), ln.weight, ln.bias, ln.eps)
    return y, (x, mean, rstd)


def _add_ln_fwd(ln, x, y):
    """``LayerNorm(x + y)`` in one fused pass, keeping what the hand-written backward needs: ``(out, (x + y, mean, rstd))``."""
    s, out, mean, rstd = ops.add_layernorm(x, y, ln.weight, ln.bias, ln.eps)
    return out, (s, mean, rstd)


def _ln_bwd(ln, saved, dy, d_res=None):
    """``LN'(dy) [+ d_res]`` for K upstream gradients against ONE forward's statistics (``ops.layernorm_bwd_add``)."""
    x, mean, rstd = saved
    return ops.layernorm_bwd_add(dy, x, mean, rstd, ln.weight, d_res)


class _FeedForward:
    """linear1 -> activation -> linear2 shared by both layer kinds (mixin; the Linear modules live on the layer)."""

    def _ffn(self, x):
        return self.linear2(self.activation(self.linear1(x)))

    def _ffn_fwd_shared(self, x):
        if self.activation is not F.relu:
            raise NotImplementedError("the shared-forward backward is written for DETR's ReLU feed-forward")
        h = self.linear1(x)
        return self.linear2(F.relu(h)), h

    def _ffn_bwd_shared(self, h, d_out, gemm_dtype=torch.float32):
        """``d_out [K, T, E]`` -> gradient w.r.t. the feed-forward input; ``h [1, T, F]`` is the shared pre-activation."""
        d_h = ops.backward_gemm(d_out, self.linear2.weight, gemm_dtype) * (h > 0)
        return ops.backward_gemm(d_h, self.linear1.weight, gemm_dtype)

    def _ffn_bwd_shared_into(self, live, d_out):
        """fp32: ``d_out [K, T, E]`` becomes ``d_out + FFN'(d_out)`` IN PLACE (the residual add is the second GEMM's beta = 1);
        ``live [1, T, F]``: the ReLU's 0 / 1 mask of the shared forward."""
        K, T, E = d_out.shape
        d_h = F.linear(d_out, ops.transposed_weight(self.linear2.weight)).mul_(live)      # (NT layout: ops.transposed_weight)
        d_out.view(K * T, E).addmm_(d_h.view(K * T, -1), self.linear1.weight)
        return d_out


class TransformerEncoderLayer(nn.Module, _FeedForward):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", normalize_before=False):
        super().__init__()
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.activation = _activation(activation)
        self.normalize_before = normalize_before

    def forward(self, src, src_mask=None, src_key_padding_mask=None, pos=None):
        if self.normalize_before:          # transformer.py forward_pre
            h = self.norm1(src)
            qk = _with_pos(h, pos)
            src = src + self.self_attn(qk, qk, h, attn_mask=src_mask, key_padding_mask=src_key_padding_mask)
            return src + self._ffn(self.norm2(src))
        qk = _with_pos(src, pos)           # transformer.py:236-256 forward_post
        a = self.self_attn(qk, qk, src, attn_mask=src_mask, key_padding_mask=src_key_padding_mask)
        src1 = self.norm1(src + a)
        r = self.activation(self.linear1(src1))
        ff = self.linear2(r)
        self._lrp = _detached(src, qk, a, src1, r, ff)       # layer inputs of the LRP pass (references, grad mode only)
        return self.norm2(src1 + ff)

    def relprop(self, cam, alpha=1, **kwargs):
        """``forward_post_relprop`` (DETR/models/transformer.py:256-275); LayerNorm / ReLU / dropout / WithPosEmbd pass
        relevance through unchanged (DETR/modules/layers.py:46-47, 110-111)."""
        src, qk, a, src1, r, ff = _lrp_tape(self)
        cam_src_2, cam_ff = lrp.add_relprop(cam, src1, ff)                               # add2([src_2, src2])
        cam_1 = lrp.linear_relprop(cam_ff, r, self.linear2.weight, alpha)
        cam_1 = lrp.linear_relprop(cam_1, src1, self.linear1.weight, alpha)
        cam = lrp.clone_relprop([cam_1, cam_src_2], src1)                                # clone3
        cam_src_3, cam_drop = lrp.add_relprop(cam, src, a)                               # add1([src_3, src_drop])
        cam_q, cam_k, cam_v = self.self_attn.relprop(cam_drop, alpha, **kwargs)
        cam_w = lrp.clone_relprop([cam_q, cam_k], qk)                                    # clone2 (X = src + pos)
        return lrp.clone_relprop([cam_w, cam_v, cam_src_3], src)                         # clone1

    # ---- shared-forward mode (post-norm): tensors are batch-first, ``[1, N, E]`` forward / ``[K, N, E]`` backward
    def forward_shared(self, src, pos, batch):
        if self.normalize_before:
            raise NotImplementedError("shared-forward mode covers the post-norm layers DETR ships")
        qk = _with_pos(src, pos)
        a, att = self.self_attn.forward_shared(qk, qk, src, batch)               # q and k: one packed GEMM
        src1, ln1 = _add_ln_fwd(self.norm1, src, a)
        ff, h = self._ffn_fwd_shared(src1)
        out, ln2 = _add_ln_fwd(self.norm2, src1, ff)
        return out, (att, ln1, h, ln2, (h > 0).to(h.dtype))

    def backward_shared(self, tape, d_out, need_input_grad=True, gemm_dtype=torch.float32):
        att, ln1, h, ln2, live = tape
        d_z2 = _ln_bwd(self.norm2, ln2, d_out)                           # w.r.t. src1 + ff
        if gemm_dtype != torch.float32:                                  # opt-in bf16 input-gradient GEMMs: the unfused chain
            d_z1 = _ln_bwd(self.norm1, ln1, d_z2 + self._ffn_bwd_shared(h, d_z2, gemm_dtype))
            dq, dk, dv = self.self_attn.backward_shared(att, d_z1, need_input_grad, gemm_dtype)
            return d_z1 + dq + dk + dv if need_input_grad else None
        d_z1 = _ln_bwd(self.norm1, ln1, self._ffn_bwd_shared_into(live, d_z2))   # w.r.t. src + attention output
        if not need_input_grad:
            self.self_attn.backward_shared(att, d_z1, False)
            return None
        # pos is a constant: q, k and v all lead to src -- d_src = d_z1 + [dq | dk | dv] . [Wq; Wk; Wv], one GEMM, in place
        return self.self_attn.backward_shared_into(att, d_z1, d_z1)


class TransformerDecoderLayer(nn.Module, _FeedForward):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", normalize_before=False):
        super().__init__()
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.activation = _activation(activation)
        self.normalize_before = normalize_before

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None, pos=None, query_pos=None):
        if self.normalize_before:          # transformer.py forward_pre
            h = self.norm1(tgt)
            qk = _with_pos(h, query_pos)
            tgt = tgt + self.self_attn(qk, qk, h, attn_mask=tgt_mask, key_padding_mask=tgt_key_padding_mask)
            h = self.norm2(tgt)
            tgt = tgt + self.multihead_attn(_with_pos(h, query_pos), _with_pos(memory, pos), memory,
                                            attn_mask=memory_mask, key_padding_mask=memory_key_padding_mask)
            return tgt + self._ffn(self.norm3(tgt))
        qk = _with_pos(tgt, query_pos)     # transformer.py:371-407 forward_post
        a = self.self_attn(qk, qk, tgt, attn_mask=tgt_mask, key_padding_mask=tgt_key_padding_mask)
        tgt1 = self.norm1(tgt + a)
        c = self.multihead_attn(_with_pos(tgt1, query_pos), _with_pos(memory, pos), memory, attn_mask=memory_mask,
                                key_padding_mask=memory_key_padding_mask)
        tgt2 = self.norm2(tgt1 + c)
        r = self.activation(self.linear1(tgt2))
        ff = self.linear2(r)
        self._lrp = _detached(tgt, qk, a, tgt1, memory, c, tgt2, r, ff)
        return self.norm3(tgt2 + ff)

    def relprop(self, cam, alpha=1, **kwargs):
        """``forward_post_relprop`` (DETR/models/transformer.py:410-436) -> ``(cam_tgt, cam_memory)``."""
        tgt, qk, a, tgt1, memory, c, tgt2, r, ff = _lrp_tape(self)
        cam_tgt_2, cam_ff = lrp.add_relprop(cam, tgt2, ff)                               # add3([tgt_2, tgt2])
        cam_ff = lrp.linear_relprop(cam_ff, r, self.linear2.weight, alpha)
        cam_tgt_1 = lrp.linear_relprop(cam_ff, tgt2, self.linear1.weight, alpha)
        cam = lrp.clone_relprop([cam_tgt_1, cam_tgt_2], tgt2)                            # clone5
        cam_tgt_2, cam_drop = lrp.add_relprop(cam, tgt1, c)                              # add2([tgt_2, tgt_drop])
        cam_q, cam_k, cam_mem_2 = self.multihead_attn.relprop(cam_drop, alpha, **kwargs)
        cam_mem = lrp.clone_relprop([cam_k, cam_mem_2], memory)                          # clone4 (wembd3 / wembd2: identity)
        cam = lrp.clone_relprop([cam_q, cam_tgt_2], tgt1)                                # clone3
        cam_tgt_3, cam_drop = lrp.add_relprop(cam, tgt, a)                               # add1([tgt_3, tgt_drop])
        cam_q, cam_k, cam_tgt_2 = self.self_attn.relprop(cam_drop, alpha, **kwargs)
        cam_tgt_1 = lrp.clone_relprop([cam_q, cam_k], qk)                                # clone2 (X = tgt + query_pos)
        return cam_tgt_1 + cam_tgt_2 + cam_tgt_3, cam_mem                                # transformer.py:434: a plain sum


    # ---- shared-forward mode (post-norm), batch-first tensors
    def forward_shared(self, tgt, memory, pos, query_pos, batch, memory_kv=None):
        """``memory_kv``: this layer's ``(k, v)`` projections of ``memory + pos`` / ``memory`` when the caller made them for all
        layers at once (``Transformer.forward_shared``)."""
        if self.normalize_before:
            raise NotImplementedError("shared-forward mode covers the post-norm layers DETR ships")
        qk = _with_pos(tgt, query_pos)
        a, t_self = self.self_attn.forward_shared(qk, qk, tgt, batch)
        tgt1, ln1 = _add_ln_fwd(self.norm1, tgt, a)
        if memory_kv is None:
            c, t_cross = self.multihead_attn.forward_shared(_with_pos(tgt1, query_pos), _with_pos(memory, pos), memory, batch)
        else:
            c, t_cross = self.multihead_attn.forward_shared(_with_pos(tgt1, query_pos), None, None, batch, kv=memory_kv)
        tgt2, ln2 = _add_ln_fwd(self.norm2, tgt1, c)
        ff, h = self._ffn_fwd_shared(tgt2)
        out, ln3 = _add_ln_fwd(self.norm3, tgt2, ff)
        return out, (t_self, ln1, t_cross, ln2, h, ln3, (h > 0).to(h.dtype))

    def backward_shared(self, tape, d_out, need_tgt_grad=True, gemm_dtype=torch.float32, d_memory=None):
        """``d_out [K, Q, E]`` -> ``(d_tgt [K, Q, E] | None, d_memory [K, N, E])``.  fp32: the memory gradient ACCUMULATES into
        ``d_memory`` (in place) when one is handed in; the opt-in bf16 route returns this layer's share."""
        t_self, ln1, t_cross, ln2, h, ln3, live = tape
        d_z3 = _ln_bwd(self.norm3, ln3, d_out)                             # w.r.t. tgt2 + ff
        if gemm_dtype != torch.float32:
            d_z2 = _ln_bwd(self.norm2, ln2, d_z3 + self._ffn_bwd_shared(h, d_z3, gemm_dtype))
            dq, dk, dv = self.multihead_attn.backward_shared(t_cross, d_z2, True, gemm_dtype)
            d_z1 = _ln_bwd(self.norm1, ln1, d_z2 + dq)
            sq, sk, sv = self.self_attn.backward_shared(t_self, d_z1, need_tgt_grad, gemm_dtype)
            share = dk + dv
            return (d_z1 + sq + sk + sv if need_tgt_grad else None), share if d_memory is None else d_memory + share
        d_z2 = _ln_bwd(self.norm2, ln2, self._ffn_bwd_shared_into(live, d_z3))          # w.r.t. tgt1 + cross-attn output
        # d_z2 <- d_z2 + dq . Wq (in place: it is dead afterwards); d_memory (+)= [dk | dv] . [Wk; Wv]
        d_sum, d_memory = self.multihead_attn.backward_shared_into(t_cross, d_z2, d_z2, d_memory, same_source=False)
        d_z1 = _ln_bwd(self.norm1, ln1, d_sum)                             # w.r.t. tgt + self-attention output
        if not need_tgt_grad:
            self.self_attn.backward_shared(t_self, d_z1, False)
            return None, d_memory
        return self.self_attn.backward_shared_into(t_self, d_z1, d_z1), d_memory


class TransformerEncoder(nn.Module):
    def __init__(self, make_layer, num_layers, norm=None):
        super().__init__()
        self.layers = nn.ModuleList([make_layer() for _ in range(num_layers)])
        self.num_layers = num_layers
        self.norm = norm

    def forward(self, src, mask=None, src_key_padding_mask=None, pos=None):
        for layer in self.layers:
            src = layer(src, src_mask=mask, src_key_padding_mask=src_key_padding_mask, pos=pos)
        return src if self.norm is None else self.norm(src)

    def relprop(self, cam, alpha=1, **kwargs):
        """DETR/models/transformer.py:104-111 (the final norm, if any, passes relevance through)."""
        for layer in reversed(self.layers):
            cam = layer.relprop(cam, alpha, **kwargs)
        return cam


class TransformerDecoder(nn.Module):
    def __init__(self, make_layer, num_layers, norm=None, return_intermediate=False):
        super().__init__()
        self.layers = nn.ModuleList([make_layer() for _ in range(num_layers)])
        self.num_layers = num_layers
        self.norm = norm
        self.return_intermediate = return_intermediate

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None, pos=None, query_pos=None):
        out, stack, outs = tgt, [], []
        for layer in self.layers:
            out = layer(out, memory, tgt_mask=tgt_mask, memory_mask=memory_mask,
                        tgt_key_padding_mask=tgt_key_padding_mask,
                        memory_key_padding_mask=memory_key_padding_mask, pos=pos, query_pos=query_pos)
            outs.append(out)
            if self.return_intermediate:
                stack.append(self.norm(out))     # transformer.py:148-153: every level is normed by the shared LN
        self._lrp = _detached(memory, *outs)
        if self.return_intermediate:
            return torch.stack(stack)
        return (out if self.norm is None else self.norm(out)).unsqueeze(0)

    def relprop(self, cam_list, alpha=1, **kwargs):
        """DETR/models/transformer.py:166-199 for ``return_intermediate=True`` (what DETR builds): ``cam_list [levels, Q, B,
        C]`` (relevance of every normed level) -> ``(cam_tgt, cam_memory)``.  Level ``j < last`` joins the stream through the
        ``Clone`` that fed both the next layer and that level's norm (``clone_list[j]``)."""
        if not self.return_intermediate:
            raise NotImplementedError("relprop covers return_intermediate_dec=True (the reference's other branch is marked "
                                      "FIXME there, transformer.py:167-172)")
        memory, *outs = _lrp_tape(self)
        cam, cam_mems = None, []
        for j in range(self.num_layers - 1, -1, -1):
            cam = cam_list[j] if j == self.num_layers - 1 else lrp.clone_relprop([cam, cam_list[j]], outs[j])
            cam, cam_mem = self.layers[j].relprop(cam, alpha, **kwargs)
            cam_mems.append(cam_mem)
        return cam, lrp.clone_relprop(cam_mems, memory)


class Transformer(nn.Module):
    """``DETR/models/transformer.py:20-64``: ``forward(src [B,C,h,w], mask [B,h,w], query_embed [Q,C], pos_embed
    [B,C,h,w]) -> (hs [levels, B, Q, C], memory [B, C, h, w])``."""

    def __init__(self, d_model=512, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=2048,
                 dropout=0.1, activation="relu", normalize_before=False, return_intermediate_dec=False):
        super().__init__()
        args = (d_model, nhead, dim_feedforward, dropout, activation, normalize_before)
        self.encoder = TransformerEncoder(lambda: TransformerEncoderLayer(*args), num_encoder_layers,
                                          nn.LayerNorm(d_model) if normalize_before else None)
        self.decoder = TransformerDecoder(lambda: TransformerDecoderLayer(*args), num_decoder_layers,
                                          nn.LayerNorm(d_model), return_intermediate=return_intermediate_dec)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        self.d_model, self.nhead = d_model, nhead

    def forward(self, src, mask, query_embed, pos_embed):
        bs, c, h, w = src.shape
        tokens = src.flatten(2).permute(2, 0, 1)                       # [hw, B, C]
        pos = pos_embed.flatten(2).permute(2, 0, 1)
        query_pos = query_embed.unsqueeze(1).expand(-1, bs, -1)
        key_padding = None if mask is None else mask.flatten(1)
        memory = self.encoder(tokens, src_key_padding_mask=key_padding, pos=pos)
        hs = self.decoder(torch.zeros_like(query_pos), memory, memory_key_padding_mask=key_padding, pos=pos,
                          query_pos=query_pos)
        self._lrp = _detached(memory)
        self._lrp_src_shape = (bs, c, h, w)
        return hs.transpose(1, 2), memory.permute(1, 2, 0).reshape(bs, c, h, w)

    def relprop(self, cam, alpha=1, **kwargs):
        """DETR/models/transformer.py:68-79: ``cam = [cam_hs [levels, B, Q, C], cam_memory [B, C, h, w]]`` -> relevance of the
        projected feature map ``[B, C, h, w]``."""
        (memory,) = _lrp_tape(self)
        bs, c, h, w = self._lrp_src_shape
        cam_hs = cam[0].transpose(1, 2)
        cam_mem1 = cam[1].reshape(bs, c, h * w).permute(2, 0, 1)
        cam_tgt, cam_mem2 = self.decoder.relprop(cam_hs, alpha, **kwargs)
        cam_memory = lrp.clone_relprop([cam_mem1, cam_mem2], memory)
        cam_src = self.encoder.relprop(cam_memory, alpha, **kwargs)
        return cam_src.permute(1, 2, 0).reshape(bs, c, h, w)


    # ---- shared-forward mode: ONE forward at batch 1, the backward at batch K (K upstream gradients)
    def forward_shared(self, src, query_embed, pos_embed, batch):
        """``src [1, C, h, w]`` -> ``(hs_last [1, Q, C], tape)``: the last decoder level through the shared decoder norm
        (what ``pred_logits`` reads).  Every attention block keeps ONE probability slab and a ``batch``-sized gradient slab.
        The decoder's cross-attentions all read the same memory: their key / value projections are TWO GEMMs for all layers
        (``memory + pos`` against the stacked ``Wk``, ``memory`` against the stacked ``Wv``)."""
        from .bert_tape import packed_linear
        tokens = src.flatten(2).transpose(1, 2)                         # [1, hw, C], batch-first
        pos = pos_embed.flatten(2).transpose(1, 2)
        query_pos = query_embed.unsqueeze(0)
        enc_tapes, dec_tapes = [], []
        memory = tokens
        for layer in self.encoder.layers:
            memory, t = layer.forward_shared(memory, pos, batch)
            enc_tapes.append(t)
        if self.encoder.norm is not None:
            raise NotImplementedError("shared-forward mode covers the post-norm encoder (no final encoder norm)")
        L = len(self.decoder.layers)
        S, E = memory.shape[1], memory.shape[2]
        H = self.decoder.layers[0].multihead_attn.num_heads
        Wk, bk = packed_linear(tuple(layer.multihead_attn.k_proj for layer in self.decoder.layers))
        Wv, bv = packed_linear(tuple(layer.multihead_attn.v_proj for layer in self.decoder.layers))
        k_all = torch.addmm(bk, (memory + pos).view(S, E), Wk.t()).view(1, S, L, H, E // H)
        v_all = torch.addmm(bv, memory.view(S, E), Wv.t()).view(1, S, L, H, E // H)
        out = torch.zeros_like(query_pos)
        for i, layer in enumerate(self.decoder.layers):
            out, t = layer.forward_shared(out, memory, pos, query_pos, batch, memory_kv=(k_all[:, :, i], v_all[:, :, i]))
            dec_tapes.append(t)
        hs, ln = _ln_fwd(self.decoder.norm, out)
        return hs, (enc_tapes, dec_tapes, ln)

    @torch.no_grad()
    def backward_shared(self, tape, d_hs, hooks=None):
        """``d_hs [K, Q, C]``: per-sample upstream gradients of ``hs_last``; fills every block's gradient slab.
        ``hooks``: ``{"decoder_done": f(), "encoder_layer_done": f(i)}`` called right after the named gradient slabs are
        complete on the current stream (the rule kernels that only need those can start beside the rest of the backward)."""
        enc_tapes, dec_tapes, ln = tape
        hooks = hooks or {}
        gd = getattr(self, "backward_gemm_dtype", torch.float32)      # torch.bfloat16: opt-in bf16 MFMA for the dX GEMMs
        d_out = _ln_bwd(self.decoder.norm, ln, d_hs)
        d_memory = None
        for i in range(len(self.decoder.layers) - 1, -1, -1):
            d_out, d_memory = self.decoder.layers[i].backward_shared(dec_tapes[i], d_out, need_tgt_grad=i > 0, gemm_dtype=gd,
                                                                     d_memory=d_memory)
        if "decoder_done" in hooks:
            hooks["decoder_done"]()
        for i in range(len(self.encoder.layers) - 1, -1, -1):         # the first layer's input is a constant of the pass
            d_memory = self.encoder.layers[i].backward_shared(enc_tapes[i], d_memory, need_input_grad=i > 0, gemm_dtype=gd)
            if "encoder_layer_done" in hooks:
                hooks["encoder_layer_done"](i)


class PositionEmbeddingSine(nn.Module):
    """``DETR/models/position_encoding.py:12-48`` on a bare ``[B, h, w]`` padding mask."""

    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and not normalize:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats, self.temperature, self.normalize = num_pos_feats, temperature, normalize
        self.scale = 2 * math.pi if scale is None else scale

    def forward(self, mask):
        valid = ~mask
        y = valid.cumsum(1, dtype=torch.float32)
        x = valid.cumsum(2, dtype=torch.float32)
        if self.normalize:
            y = y / (y[:, -1:, :] + 1e-6) * self.scale
            x = x / (x[:, :, -1:] + 1e-6) * self.scale
        i = torch.arange(self.num_pos_feats, dtype=torch.float32, device=mask.device)
        freq = self.temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / self.num_pos_feats)

        def interleave(t):
            t = t[:, :, :, None] / freq
            return torch.stack((t[..., 0::2].sin(), t[..., 1::2].cos()), dim=4).flatten(3)

        return torch.cat((interleave(y), interleave(x)), dim=3).permute(0, 3, 1, 2)


class MLP(nn.Module):
    """``DETR/models/detr.py`` ``MLP``: ``num_layers`` Linear layers with ReLU in between (the box head)."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.num_layers = num_layers
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = layer(x) if i == self.num_layers - 1 else F.relu(layer(x))
        return x


class DETRFromFeatures(nn.Module):
    """Everything of ``DETR.forward`` (``detr.py:43-70``) after the backbone.

    ``model(features [B, C_backbone, h, w], mask [B, h, w] bool | None) -> {'pred_logits': [B, Q, classes+1],
    'pred_boxes': [B, Q, 4]}`` -- the duck type ``detr_explainability.Generator`` drives.
    """

    def __init__(self, transformer, num_classes, num_queries, backbone_channels=2048, pos_normalize=True):
        super().__init__()
        d = transformer.d_model
        self.num_queries = num_queries
        self.transformer = transformer
        self.class_embed = nn.Linear(d, num_classes + 1)
        self.bbox_embed = MLP(d, d, 4, 3)
        self.query_embed = nn.Embedding(num_queries, d)
        self.input_proj = nn.Conv2d(backbone_channels, d, kernel_size=1)
        self.position = PositionEmbeddingSine(d // 2, normalize=pos_normalize)

    def _unpadded(self, features):
        """``(mask, position embedding)`` of an unpadded feature map, computed once per (batch, h, w, device): the same
        tensors for every image of that size (and nothing to compute inside a hipGraph capture)."""
        key = (features.shape[0], features.shape[-2], features.shape[-1], features.device)
        cache = self.__dict__.setdefault("_pos_cache", {})
        if key not in cache:
            mask = torch.zeros(features.shape[0], *features.shape[-2:], dtype=torch.bool, device=features.device)
            with torch.no_grad():
                cache[key] = (mask, self.position(mask))
        return cache[key]

    def forward(self, features, mask=None):
        if mask is None:
            mask, pos = self._unpadded(features)
        else:
            pos = self.position(mask)
        self.spatial_dim = features.shape[-2:]
        # 1x1 convolution == per-pixel Linear: run it as a GEMM (MIOpen's generic conv path is far slower on gfx950)
        proj = F.linear(features.permute(0, 2, 3, 1), self.input_proj.weight.flatten(1), self.input_proj.bias)
        hs, memory = self.transformer(proj.permute(0, 3, 1, 2), mask, self.query_embed.weight, pos)
        self.memory_shape = memory.shape
        logits = self.class_embed(hs[-1])
        self._lrp = _detached(hs, logits)
        return {"pred_logits": logits, "pred_boxes": self.bbox_embed(hs[-1]).sigmoid()}

    @torch.no_grad()
    def relprop(self, cam=None, alpha=1, **kwargs):
        """``DETR.relprop`` (DETR/models/detr.py:79-92): the LRP pass of the whole head.  ``target_index`` (kept queries) and
        ``target_class`` (``None``: arg-max over all classes, as there) select the relevance seeds; ``cam`` is ignored, as in
        the reference.  Afterwards every attention module holds its ``attn_cam`` (``save_attn_cam``).  Returns the relevance
        of the projected feature map ``[B, C, h, w]`` (the reference stops there too).  The decoder level that feeds
        ``pred_logits`` is the last one here (the reference hard-codes ``index_select(..., [5])`` of its 6 levels)."""
        hs, logits = _lrp_tape(self)
        if logits.shape[0] != 1:
            # the reference explains item 0 of a one-image batch (detr.py:85-86 seeds ``one_hot[0, ...]``) and its Linear / Add / MHA
            # rules renormalise with WHOLE-TENSOR sums (layers.py:409-437): on a batch these would mix the samples silently
            raise NotImplementedError("DETR relprop runs on a one-image batch, like the reference's generators (got batch %d): "
                                      "call it per image" % logits.shape[0])
        target_index, target_class = kwargs["target_index"], kwargs.get("target_class")
        if target_class is None:
            target_class = logits.max(dim=-1)[1][0, target_index]
        seed = torch.zeros_like(logits)
        seed[0, target_index, target_class] = 1
        # index_select.relprop (DETR/modules/layers.py:232-244): only the selected level carries relevance -- X * (R / X)
        cam_logits = logits * lrp.safe_divide(seed, logits)
        levels_logits = torch.zeros(hs.shape[0], *logits.shape, dtype=logits.dtype, device=logits.device)
        levels_logits[-1] = cam_logits
        # class_embed.relprop on ALL levels at once, as the reference does (its R.sum() / Z cover the whole stack)
        cam_hs = lrp.linear_relprop(levels_logits, hs, self.class_embed.weight, alpha)
        mem_zero = torch.zeros(self.memory_shape, dtype=hs.dtype, device=hs.device)
        return self.transformer.relprop([cam_hs, mem_zero], alpha, **kwargs)


    def forward_shared(self, features, batch, mask=None):
        """Shared-forward mode for ``Generator.generate_ours_multi``: ``features [1, C, h, w]`` ->
        ``(pred_logits [1, Q, classes+1], state)``; see ``Transformer.forward_shared``."""
        if features.shape[0] != 1:
            raise ValueError("forward_shared takes the single shared image")
        pos = self._unpadded(features)[1] if mask is None else self.position(mask)
        self.spatial_dim = features.shape[-2:]
        proj = F.linear(features.permute(0, 2, 3, 1), self.input_proj.weight.flatten(1), self.input_proj.bias)
        hs, tape = self.transformer.forward_shared(proj.permute(0, 3, 1, 2), self.query_embed.weight, pos, batch)
        return self.class_embed(hs), tape

    @torch.no_grad()
    def backward_shared(self, state, d_logits, hooks=None):
        """``d_logits [K, Q, classes+1]``: one upstream gradient of ``pred_logits`` per explained target.  ``hooks``: see
        ``Transformer.backward_shared``."""
        self.transformer.backward_shared(state, torch.matmul(d_logits, self.class_embed.weight), hooks)


def detr_resnet50_head(num_classes=91, num_queries=100):
    """The transformer + heads of DETR-R50 (``DETR/main.py`` defaults: d=256, 8 heads, 6+6 layers, ffn 2048)."""
    return DETRFromFeatures(Transformer(d_model=256, nhead=8, num_encoder_layers=6, num_decoder_layers=6,
                                        dim_feedforward=2048, dropout=0.1, return_intermediate_dec=True),
                            num_classes=num_classes, num_queries=num_queries)
