"""CPU oracle (torch fp32, autograd) of the reference's LXMERT body under ``GeneratorOurs.generate_ours``.

TEST INFRASTRUCTURE ONLY (see ``oracle/relevancy_np.py`` header for who may import this).  Self-contained: takes a plain
``state_dict`` with the reference's parameter names (``lxmert.embeddings.*``, ``lxmert.encoder.{visn_fc,layer,r_layers,
x_layers}.*``, ``lxmert.pooler.*``, ``answer_head.logit_fc.*``) and runs stock torch CPU ops; it imports neither the
product package nor ``/root/reference``.

Restates, with citations (relative to /root/reference/lxmert/lxmert/src):
  * embeddings -- lxmert_lrp.py:285-310 (token type + position + word, LayerNorm eps 1e-12);
  * the visual feature encoder -- :758-767 (``(LN(fc(feat)) + LN(fc(box))) / 2``);
  * the hooked attention -- :385-420 (``scores / sqrt(d)``, additive mask, softmax; the probabilities go to ``save_attn`` and
    their gradient to ``save_attn_gradients``): ``oracle/attention_torch.core`` in ``SCALE_SCORES`` mode;
  * attention output / intermediate (exact GELU, ``layers.py:71``) / output blocks -- :472-477, :549-552, :568-573;
  * single-modality layers :592-601 and the cross-modality layer :630-664 (``visual_attention_copy`` is a deep copy made at the
    first call, i.e. the SAME weights as ``visual_attention``), :666-676, :678-690, :701-730; the encoder's layer order
    :812-842 (language layers, then relational layers, then cross layers);
  * pooler :876-884 (first token, dense, tanh) and answer head :941-953 (Linear, GELU, LayerNorm, Linear);
  * ``LxmertModel.forward``'s mask extension ``(1 - mask) * -10000`` -- :1188-1225 (no visual mask in the evaluator);
  * ``GeneratorOurs.generate_ours(use_lrp=False)`` -- ExplanationGenerator.py:131-211: forward, one-hot on the arg-max answer,
    ONE backward, then the rule schedule (``oracle/relevancy_np.lxmert_generate_ours_chain``).
Pinned by ``tests/test_oracle_golden.py::test_lxmert_torch_oracle`` against ``tests/golden/lxmert_model.npz`` (the reference's own
layers + generator, made by ``tests/golden/make_golden.py::gen_lxmert_model``).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import attention_torch as at
from . import relevancy_np as rn

EPS = 1e-12


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + "weight"], sd[p + "bias"], EPS)


def _lin(sd, p, x):
    return F.linear(x, sd[p + "weight"], sd[p + "bias"])


def _attention(sd, p, hidden, context, mask, heads, captured):
    """lxmert_lrp.py:385-420 on ``[B, N, E]``; appends the graph-attached ``P [B, H, Nq, Nk]``."""
    B, Nq, E = hidden.shape
    d = E // heads
    split = lambda t: t.view(B, -1, heads, d).permute(0, 2, 1, 3)                        # transpose_for_scores
    q, k, v = split(_lin(sd, p + "query.", hidden)), split(_lin(sd, p + "key.", context)), split(_lin(sd, p + "value.", context))
    prob, o = at.core(q, k, v, math.sqrt(d), at.SCALE_SCORES, mask)
    captured.append(prob)
    return o.permute(0, 2, 1, 3).contiguous().view(B, Nq, E)


def _att_output(sd, p, hidden, residual):                                                 # :472-477 / :568-573
    return _ln(sd, p + "LayerNorm.", _lin(sd, p + "dense.", hidden) + residual)


def _self_layer(sd, p, x, mask, heads, captured):                                         # :520-526
    return _att_output(sd, p + "output.", _attention(sd, p + "self.", x, x, mask, heads, captured), x)


def _ffn(sd, p_inter, p_out, x):
    return _att_output(sd, p_out, F.gelu(_lin(sd, p_inter + "dense.", x)), x)


def _count(sd, prefix):
    return len({k[len(prefix):].split(".")[0] for k in sd if isinstance(k, str) and k.startswith(prefix)})


def forward(sd, heads, input_ids, visual_feats, visual_pos, attention_mask=None, token_type_ids=None):
    """-> ``(score [B, labels], caps)``; ``caps``: dict of lists of graph-attached probabilities ``[B, H, Nq, Nk]`` --
    ``lang`` / ``vis`` (single-modality layers) and per cross layer ``lang_cross``, ``img_cross``, ``lang_self``, ``img_self``."""
    B, T = input_ids.shape
    if attention_mask is None:
        attention_mask = torch.ones(B, T)
    if token_type_ids is None:
        token_type_ids = torch.zeros(B, T, dtype=torch.long)
    ext = (1.0 - attention_mask[:, None, None, :].float()) * -10000.0                     # :1188-1199
    e = "lxmert.embeddings."
    pos_ids = torch.arange(T).unsqueeze(0).expand(B, T)
    lang = F.embedding(token_type_ids, sd[e + "token_type_embeddings.weight"]) \
        + F.embedding(pos_ids, sd[e + "position_embeddings.weight"])                      # add1, :306
    lang = _ln(sd, e + "LayerNorm.", lang + F.embedding(input_ids, sd[e + "word_embeddings.weight"]))
    v = "lxmert.encoder.visn_fc."
    vis = (_ln(sd, v + "visn_layer_norm.", _lin(sd, v + "visn_fc.", visual_feats))
           + _ln(sd, v + "box_layer_norm.", _lin(sd, v + "box_fc.", visual_pos))) / 2    # :758-764
    caps = dict(lang=[], vis=[], lang_cross=[], img_cross=[], lang_self=[], img_self=[])
    for l in range(_count(sd, "lxmert.encoder.layer.")):                                  # :812-818
        p = "lxmert.encoder.layer.%d." % l
        lang = _self_layer(sd, p + "attention.", lang, ext, heads, caps["lang"])
        lang = _ffn(sd, p + "intermediate.", p + "output.", lang)
    for l in range(_count(sd, "lxmert.encoder.r_layers.")):                               # :821-826
        p = "lxmert.encoder.r_layers.%d." % l
        vis = _self_layer(sd, p + "attention.", vis, None, heads, caps["vis"])
        vis = _ffn(sd, p + "intermediate.", p + "output.", vis)
    for l in range(_count(sd, "lxmert.encoder.x_layers.")):                               # :829-842
        p = "lxmert.encoder.x_layers.%d." % l
        xa = p + "visual_attention."
        lang_x = _att_output(sd, xa + "output.", _attention(sd, xa + "att.", lang, vis, None, heads, caps["lang_cross"]), lang)
        vis_x = _att_output(sd, xa + "output.", _attention(sd, xa + "att.", vis, lang, ext, heads, caps["img_cross"]), vis)
        lang_s = _self_layer(sd, p + "lang_self_att.", lang_x, ext, heads, caps["lang_self"])
        vis_s = _self_layer(sd, p + "visn_self_att.", vis_x, None, heads, caps["img_self"])
        lang = _ffn(sd, p + "lang_inter.", p + "lang_output.", lang_s)
        vis = _ffn(sd, p + "visn_inter.", p + "visn_output.", vis_s)
    pooled = torch.tanh(_lin(sd, "lxmert.pooler.dense.", lang[:, 0]))                     # :876-884
    h = F.gelu(_lin(sd, "answer_head.logit_fc.0.", pooled))                               # :941-953
    h = F.layer_norm(h, (h.shape[-1],), sd["answer_head.logit_fc.2.weight"], sd["answer_head.logit_fc.2.bias"], EPS)
    return _lin(sd, "answer_head.logit_fc.3.", h), caps


def prepare_state_dict(state_dict):
    return {k: v.detach().float().clone().requires_grad_(True) for k, v in state_dict.items()
            if torch.is_tensor(v) and v.is_floating_point()}


def generate_ours(sd, heads, inputs, index=None, normalize_self_attention=True, apply_self_in_rule_10=True, with_state=False):
    """ExplanationGenerator.py:131-211 (``use_lrp=False``) for ONE item (batch 1, as the evaluator calls it):
    ``(R_t_t [T, T], R_t_i [T, I])`` numpy fp32."""
    score, caps = forward(sd, heads, **inputs)
    if index is None:
        index = int(score[0].argmax())                                                    # :138-139
    order = ["lang", "vis", "lang_cross", "img_cross", "lang_self", "img_self"]
    flat = [t for k in order for t in caps[k]]
    # the last cross layer's image-side blocks do not reach the pooled token: zero gradient, like an unfired hook's stale
    # zero -- the schedule never reads them (ExplanationGenerator.py:196-205 stops after the language side)
    grads = torch.autograd.grad(score[0, index], flat, allow_unused=True)
    grads = [torch.zeros_like(t) if g is None else g for g, t in zip(grads, flat)]
    it = iter(grads)
    gr = {k: [next(it).numpy() for _ in caps[k]] for k in order}
    pr = {k: [t.detach().numpy() for t in caps[k]] for k in order}
    x_layers = [dict(lang_cross=(pr["lang_cross"][i], gr["lang_cross"][i]), img_cross=(pr["img_cross"][i], gr["img_cross"][i]),
                     lang_self=(pr["lang_self"][i], gr["lang_self"][i]), img_self=(pr["img_self"][i], gr["img_self"][i]))
                for i in range(len(caps["lang_cross"]))]
    R_t_t, R_t_i = rn.lxmert_generate_ours_chain(pr["lang"], gr["lang"], pr["vis"], gr["vis"], x_layers,
                                                 normalize_self_attention, apply_self_in_rule_10)
    if with_state:
        return R_t_t, R_t_i, dict(score=score.detach().numpy(), probs=pr, grads=gr)
    return R_t_t, R_t_i
