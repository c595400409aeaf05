"""CPU oracle (torch fp32, autograd) of the reference's VisualBERT body under ``SelfAttentionGenerator.generate_ours``.

TEST INFRASTRUCTURE ONLY (see ``oracle/relevancy_np.py`` header for who may import this).  Self-contained: takes a plain
``state_dict`` with the reference's parameter names (``model.bert.embeddings.*``, ``model.bert.encoder.layer.*``,
``model.classifier.*``) and runs stock torch CPU ops; it imports neither the product package nor ``/root/reference``.

Restates, with citations (relative to /root/reference/VisualBERT/mmf):
  * the visio-linguistic embedding sum -- modules/embeddings.py:325-460 (text: word + position + token type; regions:
    projection + visual position (all ids 0) + visual token type; one LayerNorm over the concatenation);
  * the hooked self-attention -- models/transformers/backends/BERT_ours.py:292-343 (``scores / sqrt(d)``, additive mask, softmax;
    probabilities to ``save_attn``, their gradient to ``save_attn_gradients``): ``oracle/attention_torch.core`` (``SCALE_SCORES``);
  * ``BertSelfOutput`` :405-409, ``BertIntermediate`` :431-434 (exact GELU), ``BertOutput`` :452-456, ``BertLayer`` :483-497;
  * the wrapper: text-padding trim and the all-ones region mask -- models/visual_bert.py:568-600; 'vqa' pooling (the token at
    ``input_mask.sum(1) - 2``) and the classifier (``BertPredictionHeadTransform`` BERT_ours.py:527-531 + Linear) -- :340-395;
  * ``SelfAttentionGenerator.generate_ours`` -- backends/ExplanationGenerator.py:68-107: one-hot on the arg-max answer, ONE
    backward, ``R += mean_h clamp(grad * cam, 0) @ R`` over the layers, the ``cls_index`` row with its own entry zeroed
    (``oracle/relevancy_np.visualbert_generate_ours_chain``).
Pinned by ``tests/test_oracle_golden.py::test_visualbert_torch_oracle`` against ``tests/golden/visualbert_model.npz`` (the
reference's own ``BERT_ours`` stack + generator, made by ``tests/golden/make_golden.py::gen_visualbert_model``).
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


def _count(sd, prefix):
    return len({k[len(prefix):].split(".")[0] for k in sd if isinstance(k, str) and k.startswith(prefix)})


def forward(sd, heads, input_ids, input_mask, image_feature_0):
    """One sample list entry (``[1, Tpad]`` ids / mask, ``[1, V, vdim]`` regions) -> ``(scores [1, labels], probs list)``."""
    n = int(input_mask.sum())                                                             # visual_bert.py:575-580 (trim)
    ids, mask = input_ids[:, :n], input_mask[:, :n]
    B, V = image_feature_0.shape[:2]
    e = "model.bert.embeddings."
    pos = torch.arange(n).unsqueeze(0).expand(B, n)
    zeros_t = torch.zeros(B, n, dtype=torch.long)
    zeros_v = torch.zeros(B, V, dtype=torch.long)
    text = F.embedding(ids, sd[e + "word_embeddings.weight"]) + F.embedding(pos, sd[e + "position_embeddings.weight"]) \
        + F.embedding(zeros_t, sd[e + "token_type_embeddings.weight"])
    vis = _lin(sd, e + "projection.", image_feature_0) + F.embedding(zeros_v, sd[e + "position_embeddings_visual.weight"]) \
        + F.embedding(zeros_v, sd[e + "token_type_embeddings_visual.weight"])
    x = _ln(sd, e + "LayerNorm.", torch.cat((text, vis), dim=1))
    attention_mask = torch.cat((mask, torch.ones(B, V, dtype=mask.dtype)), dim=-1)
    ext = (1.0 - attention_mask[:, None, None, :].float()) * -10000.0
    E = x.shape[-1]
    d = E // heads
    probs = []
    for l in range(_count(sd, "model.bert.encoder.layer.")):
        p = "model.bert.encoder.layer.%d." % l
        split = lambda t: t.view(B, -1, heads, d).permute(0, 2, 1, 3)
        a = p + "attention.self."
        prob, o = at.core(split(_lin(sd, a + "query.", x)), split(_lin(sd, a + "key.", x)), split(_lin(sd, a + "value.", x)),
                          math.sqrt(d), at.SCALE_SCORES, ext)
        probs.append(prob)
        o = o.permute(0, 2, 1, 3).contiguous().view(B, -1, E)
        x1 = _ln(sd, p + "attention.output.LayerNorm.", _lin(sd, p + "attention.output.dense.", o) + x)
        h = F.gelu(_lin(sd, p + "intermediate.dense.", x1))
        x = _ln(sd, p + "output.LayerNorm.", _lin(sd, p + "output.dense.", h) + x1)
    pooled = x[torch.arange(B), mask.sum(1) - 2]                                          # pooler_strategy == "vqa"
    c = "model.classifier."
    h = _ln(sd, c + "0.LayerNorm.", F.gelu(_lin(sd, c + "0.dense.", pooled)))
    return _lin(sd, c + "1.", h), probs


def prepare_state_dict(state_dict):
    return {k: v.detach().float().clone().requires_grad_(True) for k, v in state_dict.items()
            if torch.is_tensor(v) and v.is_floating_point()}


def generate_ours(sd, heads, input_ids, input_mask, image_feature_0, index=None, with_state=False):
    """ExplanationGenerator.py:68-107 for one item: ``[1, N]`` numpy fp32 (N = text tokens + regions)."""
    scores, probs = forward(sd, heads, input_ids, input_mask, image_feature_0)
    if index is None:
        index = int(scores[0].argmax())
    grads = torch.autograd.grad(scores[0, index], probs)
    cls_index = int(input_mask.sum()) - 2
    out = rn.visualbert_generate_ours_chain([p.detach().numpy() for p in probs], [g.numpy() for g in grads], cls_index)
    if with_state:
        return out, dict(scores=scores.detach().numpy())
    return out
