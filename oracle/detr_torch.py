"""CPU oracle (torch fp32, autograd) of the reference's DETR transformer + heads under ``Generator.generate_ours``.

TEST INFRASTRUCTURE ONLY (see ``oracle/relevancy_np.py`` header for who may import this).  Self-contained: takes a plain
``state_dict`` with the reference's parameter names (``transformer.*``, ``class_embed.*``, ``query_embed.*``,
``input_proj.*``) and runs stock torch CPU ops; it imports neither the product package nor ``/root/reference``.

Restates, with citations (relative to /root/reference):
  * the post-norm encoder / decoder layers -- DETR/models/transformer.py:230-254 (``forward_post``), :372-408, the stacks
    :89-102, :127-165 (``return_intermediate``: every level through the shared decoder norm) and ``Transformer.forward`` :51-66;
  * the hooked attention module -- DETR/modules/layers.py:727-765 (three projections, ``q * scaling``, softmax, the
    probabilities handed to ``save_attn`` and their gradient to ``save_attn_gradients``; masks are accepted and ignored
    there): ``oracle/attention_torch.core``;
  * ``DETR.forward`` after the backbone -- DETR/models/detr.py:61-70 (1x1 ``input_proj``, ``class_embed`` of the last level);
  * the sine position embedding -- DETR/models/position_encoding.py:28-48;
  * ``Generator.generate_ours(img, target_index, use_lrp=False)`` -- DETR/modules/ExplanationGenerator.py:142-195: forward,
    one-hot over ``(target_index, argmax class without the no-object column)``, ONE backward, the rule schedule
    (``oracle/relevancy_np.detr_generate_ours_chain``).  ``mask_generator.py:90-110`` calls it once per kept query:
    ``generate_ours_per_query`` is that loop (what the product's K-slot pass must reproduce row by row).
Pinned by ``tests/test_oracle_golden.py::test_detr_torch_oracle`` against ``tests/golden/detr_transformer.npz`` (the reference's
own ``transformer.py`` + ``Generator``, made by ``tests/golden/make_golden.py::gen_detr_transformer``).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

from . import attention_torch as at
from . import relevancy_np as rn


def position_embedding_sine(mask, num_pos_feats, temperature=10000, normalize=True, scale=2 * math.pi):
    """position_encoding.py:28-48 on a ``[B, h, w]`` bool padding mask -> ``[B, 2 * num_pos_feats, h, w]``."""
    not_mask = ~mask
    y_embed = not_mask.cumsum(1, dtype=torch.float32)
    x_embed = not_mask.cumsum(2, dtype=torch.float32)
    if normalize:
        y_embed = y_embed / (y_embed[:, -1:, :] + 1e-6) * scale
        x_embed = x_embed / (x_embed[:, :, -1:] + 1e-6) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / num_pos_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).flatten(3)
    pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).flatten(3)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


def _mha(sd, p, query, key, value, heads, captured):
    """layers.py:727-765, ``[T, B, E]`` / ``[S, B, E]`` sequence-first; appends the graph-attached ``P [B*H, T, S]``."""
    T, B, E = query.shape
    S = key.shape[0]
    d = E // heads
    q = F.linear(query, sd[p + "q_proj.weight"], sd[p + "q_proj.bias"])
    k = F.linear(key, sd[p + "k_proj.weight"], sd[p + "k_proj.bias"])
    v = F.linear(value, sd[p + "v_proj.weight"], sd[p + "v_proj.bias"])
    q = q.contiguous().view(T, B * heads, d).transpose(0, 1)
    k = k.contiguous().view(S, B * heads, d).transpose(0, 1)
    v = v.contiguous().view(S, B * heads, d).transpose(0, 1)
    prob, o = at.core(q, k, v, float(d) ** -0.5, at.SCALE_Q_FIRST)
    captured.append(prob)
    o = o.transpose(0, 1).contiguous().view(T, B, E)
    return F.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + "weight"], sd[p + "bias"])


def _ffn(sd, p, x):
    return F.linear(F.relu(F.linear(x, sd[p + "linear1.weight"], sd[p + "linear1.bias"])),
                    sd[p + "linear2.weight"], sd[p + "linear2.bias"])


def _count(sd, prefix):
    return len({k[len(prefix):].split(".")[0] for k in sd if isinstance(k, str) and k.startswith(prefix)})


def forward(sd, features, pos, heads):
    """``features [1, Cb, h, w]`` (backbone map), ``pos [1, d, h, w]`` -> ``(pred_logits [1, Q, classes + 1], enc, dself, dcross)``
    with the three lists holding every block's graph-attached probabilities ``[H, Nq, Nk]`` (batch 1)."""
    src = F.conv2d(features, sd["input_proj.weight"], sd["input_proj.bias"])             # detr.py:66
    bs = src.shape[0]
    x = src.flatten(2).permute(2, 0, 1)                                                  # transformer.py:54-58
    pe = pos.flatten(2).permute(2, 0, 1)
    query_pos = sd["query_embed.weight"].unsqueeze(1).repeat(1, bs, 1)
    enc, dself, dcross = [], [], []
    for l in range(_count(sd, "transformer.encoder.layers.")):                           # :230-254
        p = "transformer.encoder.layers.%d." % l
        qk = x + pe
        x = _ln(sd, p + "norm1.", x + _mha(sd, p + "self_attn.", qk, qk, x, heads, enc))
        x = _ln(sd, p + "norm2.", x + _ffn(sd, p, x))
    memory = x
    out = torch.zeros_like(query_pos)                                                    # :61
    n_dec = _count(sd, "transformer.decoder.layers.")
    for l in range(n_dec):                                                               # :372-408
        p = "transformer.decoder.layers.%d." % l
        qk = out + query_pos
        out = _ln(sd, p + "norm1.", out + _mha(sd, p + "self_attn.", qk, qk, out, heads, dself))
        out = _ln(sd, p + "norm2.", out + _mha(sd, p + "multihead_attn.", out + query_pos, memory + pe, memory, heads, dcross))
        out = _ln(sd, p + "norm3.", out + _ffn(sd, p, out))
    hs_last = _ln(sd, "transformer.decoder.norm.", out).transpose(0, 1)                  # :148-153, :66
    logits = F.linear(hs_last, sd["class_embed.weight"], sd["class_embed.bias"])         # detr.py:69-70
    return logits, enc, dself, dcross


def prepare_state_dict(state_dict):
    """fp32 leaf tensors that require grad, like ``nn.Parameter`` (the reference never freezes them)."""
    return {k: v.detach().float().clone().requires_grad_(True) for k, v in state_dict.items()
            if torch.is_tensor(v) and v.is_floating_point()}


def _np(ts):
    return [t.detach().numpy() for t in ts]


def generate_ours(sd, features, pos, target_index, heads, index=None, normalize_self_attention=True,
                  apply_self_in_rule_10=True, with_state=False):
    """ExplanationGenerator.py:142-195 with ``use_lrp=False``: ``[1, 1, len(target_index), Ni]`` (numpy fp32)."""
    logits, enc, dself, dcross = forward(sd, features, pos, heads)
    target_index = torch.as_tensor(target_index).reshape(-1)
    if index is None:
        index = logits[0, target_index, :-1].max(1)[1]                                   # :152-153
    one_hot = torch.zeros_like(logits)
    one_hot[0, target_index, index] = 1
    grads = torch.autograd.grad(torch.sum(one_hot * logits), enc + dself + dcross)       # :157-163 (one backward)
    ne, nd = len(enc), len(dself)
    g_enc, g_self, g_cross = _np(grads[:ne]), _np(grads[ne:ne + nd]), _np(grads[ne + nd:])
    out = rn.detr_generate_ours_chain(_np(enc), g_enc, _np(dself), g_self, _np(dcross), g_cross, target_index.numpy(),
                                      normalize_self_attention, apply_self_in_rule_10)
    if with_state:
        return out, dict(pred_logits=logits.detach().numpy(), enc=_np(enc), enc_grad=g_enc, dself=_np(dself), dself_grad=g_self,
                         dcross=_np(dcross), dcross_grad=g_cross)
    return out


def generate_ours_per_query(sd, features, pos, targets, heads):
    """mask_generator.py:90-110: one ``generate_ours`` call per kept query; rows stacked ``[K, Ni]``.  The forward is the same
    for every call (eval mode), so it runs once and each target gets its own backward (``retain_graph``)."""
    logits, enc, dself, dcross = forward(sd, features, pos, heads)
    ne, nd = len(enc), len(dself)
    a_enc, a_self, a_cross = _np(enc), _np(dself), _np(dcross)
    rows = []
    for t in np.asarray(targets).reshape(-1):
        t = int(t)
        cls = int(logits[0, t, :-1].argmax())
        grads = torch.autograd.grad(logits[0, t, cls], enc + dself + dcross, retain_graph=True)
        out = rn.detr_generate_ours_chain(a_enc, _np(grads[:ne]), a_self, _np(grads[ne:ne + nd]), a_cross,
                                          _np(grads[ne + nd:]), np.array([t]))
        rows.append(out[0, 0, 0])
    return np.stack(rows), logits.detach().numpy()
