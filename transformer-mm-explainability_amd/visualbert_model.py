"""VisualBERT (single-stream BERT over text tokens + image regions) on the HIP capture op -- the body
``visualbert_explainability.SelfAttentionGenerator`` drives.

Module tree and parameter names follow ``VisualBERT/mmf/models/visual_bert.py`` and
``.../transformers/backends/BERT_ours.py`` so an mmf VisualBERT-VQA2 checkpoint (``model.bert.*``,
``model.classifier.*``) loads unchanged.  Every ``encoder.layer[i].attention.self`` is
``attention_modules.BertStyleAttention``: P and dL/dP are written into device slabs by the HIP kernels and
``get_attn()`` / ``get_attn_gradients()`` (``[B, H, N, N]``) are views of them -- no hooks.

Out of scope: mmf's registry / config / dataset machinery, the pretraining and nlvr2 heads,
TorchScript, ``relprop``.  Eval mode only.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import bert_lrp, bert_tape as bt, lrp
from .attention_modules import BertStyleAttention


@dataclass
class VisualBertConfig:
    """BertConfig fields + the mmf additions this body reads (``configs/models/visual_bert/defaults.yaml``)."""
    vocab_size: int = 30522
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    max_position_embeddings: int = 512
    type_vocab_size: int = 2
    layer_norm_eps: float = 1e-12
    pad_token_id: int = 0
    hidden_act: str = "gelu"
    visual_embedding_dim: int = 2048
    num_labels: int = 3129
    pooler_strategy: str = "vqa"          # "vqa": pool the second-to-last text token (visual_bert.py:376-386)


class BertVisioLinguisticEmbeddings(nn.Module):                        # mmf/modules/embeddings.py:305-460
    def __init__(self, c):
        super().__init__()
        self.word_embeddings = nn.Embedding(c.vocab_size, c.hidden_size, padding_idx=c.pad_token_id)
        self.position_embeddings = nn.Embedding(c.max_position_embeddings, c.hidden_size)
        self.token_type_embeddings = nn.Embedding(c.type_vocab_size, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.token_type_embeddings_visual = nn.Embedding(c.type_vocab_size, c.hidden_size)
        self.position_embeddings_visual = nn.Embedding(c.max_position_embeddings, c.hidden_size)
        self.projection = nn.Linear(c.visual_embedding_dim, c.hidden_size)

    def encode_text(self, input_ids, token_type_ids=None):
        pos = torch.arange(input_ids.shape[1], device=input_ids.device).unsqueeze(0).expand_as(input_ids)
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        return (self.word_embeddings(input_ids) + self.position_embeddings(pos)
                + self.token_type_embeddings(token_type_ids))

    def get_position_embeddings_visual(self, visual_embeddings, image_text_alignment=None):
        zero_ids = torch.zeros(visual_embeddings.shape[:-1], dtype=torch.long, device=visual_embeddings.device)
        base = self.position_embeddings_visual(zero_ids)
        if image_text_alignment is None:
            return base
        # mean of the aligned words' position embeddings; -1 marks padding (embeddings.py:373-408)
        valid = (image_text_alignment != -1).long()
        aligned = (self.position_embeddings(valid * image_text_alignment) * valid.unsqueeze(-1)).sum(2)
        return aligned / valid.sum(2).clamp(min=1).unsqueeze(-1) + base

    def encode_image(self, visual_embeddings, visual_embeddings_type, image_text_alignment=None):
        projected = self.projection(visual_embeddings)
        return (projected + self.get_position_embeddings_visual(projected, image_text_alignment)
                + self.token_type_embeddings_visual(visual_embeddings_type))

    def forward(self, input_ids, token_type_ids=None, visual_embeddings=None, visual_embeddings_type=None,
                image_text_alignment=None):
        emb = self.encode_text(input_ids, token_type_ids)
        if visual_embeddings is not None and visual_embeddings_type is not None:
            emb = torch.cat((emb, self.encode_image(visual_embeddings, visual_embeddings_type,
                                                    image_text_alignment)), dim=1)
        return self.LayerNorm(emb)


class _DenseAddNorm(nn.Module):                                        # BertSelfOutput / BertOutput
    def __init__(self, c, in_features):
        super().__init__()
        self.dense = nn.Linear(in_features, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)

    def forward(self, hidden_states, input_tensor):
        d = self.dense(hidden_states)
        self._lrp_tape = None                                       # a stale tape must not outlive this forward
        if torch.is_grad_enabled():                                    # Add / Linear inputs of the LRP pass (bert_lrp.py)
            self._lrp_tape = (hidden_states.detach(), d.detach(), input_tensor.detach())
        return self.LayerNorm(d + input_tensor)


class BertAttention(nn.Module):                                        # BERT_ours.py:189-232
    def __init__(self, c):
        super().__init__()
        self.self = BertStyleAttention(c.hidden_size, c.num_attention_heads)
        self.output = _DenseAddNorm(c, c.hidden_size)

    def forward(self, hidden_states, attention_mask=None):
        return self.output(self.self(hidden_states, None, attention_mask)[0], hidden_states)


class BertIntermediate(nn.Module):                                     # BERT_ours.py:422-442
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.intermediate_size)
        self.intermediate_act_fn = {"gelu": F.gelu, "relu": F.relu}[c.hidden_act]

    def forward(self, hidden_states):
        self._lrp_tape = None                                       # a stale tape must not outlive this forward
        if torch.is_grad_enabled():
            self._lrp_tape = hidden_states.detach()
        return self.intermediate_act_fn(self.dense(hidden_states))


class BertLayer(nn.Module):                                            # BERT_ours.py:475-515
    def __init__(self, c):
        super().__init__()
        self.attention = BertAttention(c)
        self.intermediate = BertIntermediate(c)
        self.output = _DenseAddNorm(c, c.intermediate_size)

    def forward(self, hidden_states, attention_mask=None):
        attended = self.attention(hidden_states, attention_mask)
        return self.output(self.intermediate(attended), attended)

    def relprop(self, cam, **kwargs):
        return bert_lrp.bert_layer_relprop(self, cam, kwargs.get("core"))


class BertEncoder(nn.Module):                                          # BERT_ours.py:93-157
    def __init__(self, c):
        super().__init__()
        self.layer = nn.ModuleList(BertLayer(c) for _ in range(c.num_hidden_layers))

    def forward(self, hidden_states, attention_mask=None):
        for blk in self.layer:
            hidden_states = blk(hidden_states, attention_mask)
        return hidden_states

    def relprop(self, cam, **kwargs):                                  # BERT_ours.py:152-156
        for blk in reversed(self.layer):
            cam = blk.relprop(cam, **kwargs)
        return cam


class BertPooler(nn.Module):                                           # BERT_ours.py:159-187
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.hidden_size)

    def forward(self, hidden_states):
        return torch.tanh(self.dense(hidden_states[:, 0]))


class BertPredictionHeadTransform(nn.Module):                          # BERT_ours.py:517-538
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.transform_act_fn = {"gelu": F.gelu, "relu": F.relu}[c.hidden_act]

    def forward(self, hidden_states):
        self._lrp_tape = None                                       # a stale tape must not outlive this forward
        if torch.is_grad_enabled():
            self._lrp_tape = hidden_states.detach()
        return self.LayerNorm(self.transform_act_fn(self.dense(hidden_states)))

    def relprop(self, cam, **kwargs):                                  # BERT_ours.py:533-537: LayerNorm / activation pass through
        return lrp.linear_relprop(cam, bert_lrp.tape_of(self), self.dense.weight, normalize=False)


class VisualBERTBase(nn.Module):                                       # visual_bert.py:34-153 (no bypass_transformer)
    def __init__(self, c):
        super().__init__()
        self.config = c
        self.embeddings = BertVisioLinguisticEmbeddings(c)
        self.encoder = BertEncoder(c)
        self.pooler = BertPooler(c)

    def forward(self, input_ids, attention_mask=None, token_type_ids=None, visual_embeddings=None,
                visual_embeddings_type=None, image_text_alignment=None):
        emb = self.embeddings(input_ids, token_type_ids, visual_embeddings, visual_embeddings_type,
                              image_text_alignment)
        if attention_mask is None:
            attention_mask = torch.ones(emb.shape[:2], device=emb.device)
        extended = (1.0 - attention_mask[:, None, None, :].to(emb.dtype)) * -10000.0
        sequence_output = self.encoder(emb, extended)
        return sequence_output, self.pooler(sequence_output), []


class VisualBERTForClassification(nn.Module):                          # visual_bert.py:280-405
    def __init__(self, c):
        super().__init__()
        self.config = c
        self.num_labels = c.num_labels
        self.pooler_strategy = c.pooler_strategy
        self.bert = VisualBERTBase(c)
        self.classifier = nn.Sequential(BertPredictionHeadTransform(c), nn.Linear(c.hidden_size, c.num_labels))

    def forward(self, input_ids, input_mask, attention_mask=None, token_type_ids=None, visual_embeddings=None,
                visual_embeddings_type=None, image_text_alignment=None, masked_lm_labels=None):
        sequence_output, pooled_output, _ = self.bert(input_ids, attention_mask, token_type_ids, visual_embeddings,
                                                      visual_embeddings_type, image_text_alignment)
        if self.pooler_strategy == "vqa":
            index = input_mask.sum(1) - 2                              # second-to-last text token
            pooled_output = sequence_output[torch.arange(sequence_output.shape[0], device=index.device), index]
        transformed = self.classifier[0](pooled_output)
        self._lrp_tape = None                                       # a stale tape must not outlive this forward
        if torch.is_grad_enabled():
            self._lrp_tape = (sequence_output.detach(), index if self.pooler_strategy == "vqa" else None, transformed.detach())
        return {"scores": self.classifier[1](transformed).reshape(-1, self.num_labels)}

    def relprop(self, cam, **kwargs):
        """``model.relprop(one_hot, alpha=1)`` (visual_bert.py:398-403, :150-153): the classifier's two Linear rules, the
        ``IndexSelect`` of the pooled token (``vqa_pooler``), the encoder top-down (BERT_ours.py:152-156).  Fills
        ``get_attn_cam()`` of every ``BertSelfAttention``; returns the relevance of the encoder input ``[B, N, E]``.  The
        reference's pass is defined for ``pooler_strategy == "vqa"`` only (it always goes through ``vqa_pooler``)."""
        if kwargs.get("alpha", 1) != 1:
            raise NotImplementedError("the generators call relprop with alpha = 1 (ExplanationGenerator.py:27)")
        sequence_output, index, transformed = bert_lrp.tape_of(self)
        if index is None:
            raise NotImplementedError("relprop: the reference's LRP pass exists for pooler_strategy == 'vqa' only")
        with torch.no_grad():
            cam = lrp.linear_relprop(cam.to(torch.float32), transformed, self.classifier[1].weight, normalize=False)
            cam = self.classifier[0].relprop(cam, **kwargs)                                     # [B, E]
            rows = torch.arange(sequence_output.shape[0], device=index.device)
            picked = sequence_output[rows, index]
            full = torch.zeros_like(sequence_output)
            full[rows, index] = picked * lrp.safe_divide(cam, picked)      # IndexSelect.relprop, one token per sample
            return self.bert.encoder.relprop(full, **kwargs)

    # ---- tape path of the explainability pass (bert_tape.py): same scores, P of every block in its slab; ``backward_tape``
    # fills the gradient slabs from d(scores) with a hand-written vector-Jacobian chain (no autograd graph through the stack)
    def forward_tape(self, input_ids, input_mask, attention_mask=None, token_type_ids=None, visual_embeddings=None,
                     visual_embeddings_type=None, image_text_alignment=None):
        bert = self.bert
        with torch.no_grad():
            x = bert.embeddings(input_ids, token_type_ids, visual_embeddings, visual_embeddings_type, image_text_alignment)
            if attention_mask is None:
                attention_mask = torch.ones(x.shape[:2], device=x.device)
            extended = (1.0 - attention_mask[:, None, None, :].to(x.dtype)) * -10000.0
            tapes = []
            for blk in bert.encoder.layer:
                x, t = bt.layer_fwd(blk, x, extended)
                tapes.append(t)
            rows = torch.arange(x.shape[0], device=x.device)
            index = input_mask.sum(1) - 2 if self.pooler_strategy == "vqa" else torch.zeros_like(rows)
            picked = x[rows, index]
        leaf = picked.detach().requires_grad_(True)
        with torch.enable_grad():                                      # the head on B rows: a handful of tiny autograd ops
            pooled = leaf if self.pooler_strategy == "vqa" else torch.tanh(bert.pooler.dense(leaf))
            scores = self.classifier(pooled).reshape(-1, self.num_labels)
        return scores, (tapes, leaf, scores, x.shape, rows, index)

    @torch.no_grad()
    def backward_tape(self, state, d_scores):
        tapes, leaf, scores, shape, rows, index = state
        with torch.enable_grad():
            (d_leaf,) = torch.autograd.grad(scores, leaf, d_scores, retain_graph=True)
        dx = torch.zeros(shape, dtype=torch.float32, device=d_leaf.device)
        dx[rows, index] = d_leaf                                       # only the pooled token feeds the classifier
        layers = self.bert.encoder.layer
        for i in range(len(layers) - 1, -1, -1):
            dx = bt.layer_bwd(layers[i], tapes[i], dx, need_input=i > 0)


class VisualBERT(nn.Module):
    """``VisualBERT.forward(sample_list)`` (visual_bert.py:408-620, classification head): ``sample_list`` is a dict
    with ``input_ids``, ``input_mask``, ``segment_ids`` (``[B, T]``), ``image_feature_0`` (``[B, V, dim]``) and
    optionally ``image_dim`` (valid regions per sample).  Like the reference it trims the text padding (batch-1
    evaluator assumption, visual_bert.py:578-588) and rewrites those entries of ``sample_list`` in place -- the
    generator reads ``input['input_mask']`` afterwards."""

    def __init__(self, c):
        super().__init__()
        self.config = c
        self.model = VisualBERTForClassification(c)

    def forward(self, sample_list):
        feats = sample_list["image_feature_0"]
        image_dim = sample_list.get("image_dim")
        if image_dim is None:
            image_dim = torch.full((feats.shape[0], 1), feats.shape[1], device=feats.device)
        if image_dim.dim() < 2:
            image_dim = image_dim.unsqueeze(-1)
        image_mask = (torch.arange(feats.shape[1], device=feats.device).expand(feats.shape[:-1]) < image_dim).long()
        n_text = int(sample_list["input_mask"].sum())                  # one D2H read, as in the reference
        for key in ("input_ids", "input_mask", "segment_ids"):
            sample_list[key] = sample_list[key][:, :n_text]
        sample_list["token_type_ids"] = sample_list["segment_ids"]
        sample_list["visual_embeddings"] = feats
        sample_list["image_mask"] = image_mask
        sample_list["visual_embeddings_type"] = torch.zeros_like(image_mask)
        sample_list["attention_mask"] = torch.cat((sample_list["input_mask"], image_mask), dim=-1)
        return self.model(sample_list["input_ids"], sample_list["input_mask"], sample_list["attention_mask"],
                          sample_list["token_type_ids"], feats, sample_list["visual_embeddings_type"],
                          sample_list.get("image_text_alignment"))

    def relprop(self, cam, **kwargs):                                  # visual_bert.py:615-616
        return self.model.relprop(cam, **kwargs)
