"""LXMERT (two-stream VQA transformer) on the HIP capture op -- the body ``lxmert_explainability`` drives.

Module tree and parameter names follow ``lxmert/lxmert/src/lxmert_lrp.py`` (itself HuggingFace's LXMERT), so an
``unc-nlp/lxmert-vqa-uncased`` state dict loads unchanged: ``lxmert.embeddings``, ``lxmert.encoder.{visn_fc, layer[9],
r_layers[5], x_layers[5]}``, ``lxmert.pooler``, ``answer_head``.  Every attention core is
``attention_modules.BertStyleAttention``: P and dL/dP are written by the HIP kernels into device slabs, and
``get_attn()`` / ``get_attn_gradients()`` (``[B, H, Nq, Nk]``) return views of them -- no hooks.

Things a caller can observe:
  * ``x_layers[i].visual_attention_copy`` (the image->text direction of the shared cross-attention weights,
    lxmert_lrp.py:640-656) exists from construction and SHARES the projection / output modules with
    ``visual_attention`` instead of deep-copying them at the first forward; it is not part of the state dict;
  * eval mode only (dropout = identity); no ``relprop`` (LRP is out of scope);
  * the Faster R-CNN feature extractor and the tokenizer (``ModelUsage`` in the reference's perturbation script) stay
    outside: inputs are ``input_ids``, ``visual_feats [B, I, 2048]``, ``visual_pos [B, I, 4]`` and the two masks.
"""
from __future__ import annotations

import contextlib
import types
from dataclasses import dataclass

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import bert_lrp, bert_tape as bt, lrp, ops
from .attention_modules import BertStyleAttention


@dataclass
class LxmertConfig:
    """The fields of ``transformers``' LxmertConfig this body reads (any object with these attributes works)."""
    vocab_size: int = 30522
    hidden_size: int = 768
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    l_layers: int = 9
    x_layers: int = 5
    r_layers: int = 5
    visual_feat_dim: int = 2048
    visual_pos_dim: int = 4
    max_position_embeddings: int = 512
    type_vocab_size: int = 2
    num_qa_labels: int = 3129
    hidden_act: str = "gelu"


def _act(name):
    return {"gelu": F.gelu, "relu": F.relu, "tanh": torch.tanh}[name]


class LxmertEmbeddings(nn.Module):                                     # lxmert_lrp.py:268-310
    def __init__(self, c):
        super().__init__()
        self.word_embeddings = nn.Embedding(c.vocab_size, c.hidden_size, padding_idx=0)
        self.position_embeddings = nn.Embedding(c.max_position_embeddings, c.hidden_size, padding_idx=0)
        self.token_type_embeddings = nn.Embedding(c.type_vocab_size, c.hidden_size, padding_idx=0)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=1e-12)

    def forward(self, input_ids, token_type_ids=None, inputs_embeds=None):
        if inputs_embeds is None:
            inputs_embeds = self.word_embeddings(input_ids)
        n = inputs_embeds.shape[1]
        pos = torch.arange(n, device=inputs_embeds.device).unsqueeze(0).expand(inputs_embeds.shape[:2])
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(pos)
        return self.LayerNorm(self.token_type_embeddings(token_type_ids) + self.position_embeddings(pos)
                              + inputs_embeds)


class LxmertAttentionOutput(nn.Module):                                # lxmert_lrp.py:464-477 (also LxmertOutput)
    def __init__(self, c, in_features=None):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size if in_features is None else in_features, c.hidden_size)
        self.LayerNorm = nn.LayerNorm(c.hidden_size, eps=1e-12)

    def forward(self, hidden_states, input_tensor):
        d = self.dense(hidden_states)
        self._lrp_tape = None                                       # a stale tape must not outlive this forward
        if torch.is_grad_enabled():                                    # Add / Linear inputs of the LRP pass (bert_lrp.py)
            self._lrp_tape = (hidden_states.detach(), d.detach(), input_tensor.detach())
        return self.LayerNorm(d + input_tensor)


class LxmertCrossAttentionLayer(nn.Module):                            # lxmert_lrp.py:489-503
    def __init__(self, c, share_weights_with=None):
        super().__init__()
        twin = share_weights_with
        self.att = BertStyleAttention(c.hidden_size, c.num_attention_heads,
                                      share_weights_with=None if twin is None else twin.att)
        self.output = LxmertAttentionOutput(c) if twin is None else twin.output

    def forward(self, input_tensor, ctx_tensor, ctx_att_mask=None, output_attentions=False):
        out = self.att(input_tensor, ctx_tensor, ctx_att_mask, output_attentions=output_attentions)
        y = self.output(out[0], input_tensor)
        # ``output`` is ONE module for both directions of an x-layer (the reference deep-copies the layer, lxmert_lrp.py:640-641):
        # each direction keeps its own record of that call
        self._lrp_out = getattr(self.output, "_lrp_tape", None)
        return (y,) + out[1:]

    def relprop(self, cam, **kwargs):
        return bert_lrp.cross_layer_relprop(self, cam, kwargs.get("core"))


class LxmertSelfAttentionLayer(nn.Module):                             # lxmert_lrp.py:513-531
    def __init__(self, c):
        super().__init__()
        self.self = BertStyleAttention(c.hidden_size, c.num_attention_heads)
        self.output = LxmertAttentionOutput(c)

    def forward(self, input_tensor, attention_mask, output_attentions=False):
        out = self.self(input_tensor, input_tensor, attention_mask, output_attentions=output_attentions)
        y = self.output(out[0], input_tensor)
        self._lrp_out = getattr(self.output, "_lrp_tape", None)
        return (y,) + out[1:]

    def relprop(self, cam, **kwargs):
        return bert_lrp.self_layer_relprop(self, cam, kwargs.get("core"))


class LxmertIntermediate(nn.Module):                                   # lxmert_lrp.py:543-552
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.intermediate_size)
        self.intermediate_act_fn = _act(c.hidden_act)

    def forward(self, hidden_states):
        self._lrp_tape = None                                       # a stale tape must not outlive this forward
        if torch.is_grad_enabled():
            self._lrp_tape = hidden_states.detach()
        return self.intermediate_act_fn(self.dense(hidden_states))


class LxmertOutput(LxmertAttentionOutput):                             # lxmert_lrp.py:560-573
    def __init__(self, c):
        super().__init__(c, in_features=c.intermediate_size)


class LxmertLayer(nn.Module):                                          # lxmert_lrp.py:584-599
    def __init__(self, c):
        super().__init__()
        self.attention = LxmertSelfAttentionLayer(c)
        self.intermediate = LxmertIntermediate(c)
        self.output = LxmertOutput(c)

    def forward(self, hidden_states, attention_mask=None, output_attentions=False):
        out = self.attention(hidden_states, attention_mask, output_attentions=output_attentions)
        return (self.output(self.intermediate(out[0]), out[0]),) + out[1:]

    def relprop(self, cam, **kwargs):
        return bert_lrp.lxmert_layer_relprop(self, cam, kwargs.get("core"))


class LxmertXLayer(nn.Module):                                         # lxmert_lrp.py:609-740
    def __init__(self, c):
        super().__init__()
        self.visual_attention = LxmertCrossAttentionLayer(c)
        self.lang_self_att = LxmertSelfAttentionLayer(c)
        self.visn_self_att = LxmertSelfAttentionLayer(c)
        self.lang_inter = LxmertIntermediate(c)
        self.lang_output = LxmertOutput(c)
        self.visn_inter = LxmertIntermediate(c)
        self.visn_output = LxmertOutput(c)
        # second direction of the same weights; kept out of _modules so checkpoints keep the reference's key set
        object.__setattr__(self, "visual_attention_copy",
                           LxmertCrossAttentionLayer(c, share_weights_with=self.visual_attention))

    def forward(self, lang_feats, lang_attention_mask, visual_feats, visual_attention_mask, output_attentions=False):
        lang_att = self.visual_attention(lang_feats, visual_feats, ctx_att_mask=visual_attention_mask,
                                         output_attentions=output_attentions)
        visn_att = self.visual_attention_copy(visual_feats, lang_feats, ctx_att_mask=lang_attention_mask)
        lang = self.lang_self_att(lang_att[0], lang_attention_mask)[0]
        visn = self.visn_self_att(visn_att[0], visual_attention_mask)[0]
        lang = self.lang_output(self.lang_inter(lang), lang)
        visn = self.visn_output(self.visn_inter(visn), visn)
        return (lang, visn) + lang_att[1:]

    def relprop(self, cam, **kwargs):
        """lxmert_lrp.py:735-740: ``relprop_output`` (:691-700), ``relprop_self`` (:672-676), ``relprop_cross`` (:657-664)."""
        core = kwargs.get("core")
        cam_lang, cam_vis = cam
        cam_vis = bert_lrp.ffn_relprop(self.visn_inter, self.visn_output, bert_lrp.tape_of(self.visn_inter), bert_lrp.tape_of(self.visn_output),
                                       cam_vis)
        cam_lang = bert_lrp.ffn_relprop(self.lang_inter, self.lang_output, bert_lrp.tape_of(self.lang_inter),
                                        bert_lrp.tape_of(self.lang_output), cam_lang)
        cam_vis = bert_lrp.self_layer_relprop(self.visn_self_att, cam_vis, core)
        cam_lang = bert_lrp.self_layer_relprop(self.lang_self_att, cam_lang, core)
        cam_vis2, cam_lang2 = bert_lrp.cross_layer_relprop(self.visual_attention_copy, cam_vis, core)
        cam_lang1, cam_vis1 = bert_lrp.cross_layer_relprop(self.visual_attention, cam_lang, core)
        lang_in = bert_lrp.tape_of(self.visual_attention.att)["hidden"]           # the x-layer's inputs (clone1 / clone2)
        vis_in = bert_lrp.tape_of(self.visual_attention_copy.att)["hidden"]
        return lrp.clone_relprop((cam_lang1, cam_lang2), lang_in), lrp.clone_relprop((cam_vis1, cam_vis2), vis_in)

    def forward_tape(self, lang, lang_mask, visn, visn_mask, side=None):
        """Same computation as ``forward`` on the tape (``bert_tape``) -> ``(lang, visn, tape)``.

        After the two cross-attentions read both inputs, the layer is two independent chains (text: cross -> self -> FFN;
        image: the same).  ``side``: a stream for the image chain -- the caller has made both streams wait for each other,
        ``lang`` / everything returned for the text chain lives on the current stream, ``visn`` and the image chain on ``side``."""
        va, vc = self.visual_attention, self.visual_attention_copy
        side_ctx = contextlib.nullcontext() if side is None else torch.cuda.stream(side)
        if side is not None:          # each input is read on the other chain's stream too: its block must outlive those reads
            lang.record_stream(side)
            visn.record_stream(torch.cuda.current_stream())
        ctx_l, t_al = bt.attention_fwd(va.att, lang, visn, visn_mask)            # text -> image
        lang1, t_ol = bt.dense_add_norm_fwd(va.output, ctx_l, lang)
        lang2, t_sl = bt.self_block_fwd(self.lang_self_att, lang1, lang_mask)
        lang3, t_fl = bt.ffn_fwd(self.lang_inter, self.lang_output, lang2)
        with side_ctx:
            ctx_v, t_av = bt.attention_fwd(vc.att, visn, lang, lang_mask)        # image -> text (the same weights)
            visn1, t_ov = bt.dense_add_norm_fwd(vc.output, ctx_v, visn)
            visn2, t_sv = bt.self_block_fwd(self.visn_self_att, visn1, visn_mask)
            visn3, t_fv = bt.ffn_fwd(self.visn_inter, self.visn_output, visn2)
        return lang3, visn3, (t_al, t_ol, t_av, t_ov, t_sl, t_sv, t_fl, t_fv)

    def backward_tape(self, tape, d_lang3, d_visn3, side=None):
        """Gradients w.r.t. the block outputs -> gradients w.r.t. its inputs ``(d_lang, d_visn)``; ``d_visn3`` may be ``None``
        (the top x-layer: the answer reads the language stream only).  ``side``: as in ``forward_tape`` (``d_visn3`` and the
        returned ``d_visn`` live on it; the two chains meet once, for the cross terms)."""
        t_al, t_ol, t_av, t_ov, t_sl, t_sv, t_fl, t_fv = tape
        va, vc = self.visual_attention, self.visual_attention_copy
        d_lang1 = bt.self_block_bwd(self.lang_self_att, t_sl, bt.ffn_bwd(self.lang_inter, self.lang_output, t_fl, d_lang3))
        d_ctx_l, d_lang_res = bt.dense_add_norm_bwd(va.output, t_ol, d_lang1)
        if d_visn3 is None:
            # the image stream's self-attention / feed-forward of this block feed nothing: their dL/dP is zero (the reference's
            # hooks never fire for them either -- the generators do not read those two modules of the last x-layer)
            self.visn_self_att.self.get_attn_gradients().zero_()
            vc.att.get_attn_gradients().zero_()
            d_lang, d_visn = bt.attention_bwd(va.att, t_al, d_ctx_l, True, d_hidden_res=d_lang_res)
            if side is not None:
                side.wait_stream(torch.cuda.current_stream())                 # d_visn moves to the image chain's stream
                d_visn.record_stream(side)
            return d_lang, d_visn
        if side is None:
            d_visn1 = bt.self_block_bwd(self.visn_self_att, t_sv, bt.ffn_bwd(self.visn_inter, self.visn_output, t_fv, d_visn3))
            d_ctx_v, d_visn_res = bt.dense_add_norm_bwd(vc.output, t_ov, d_visn1)
            # lang1 = LN(dense(att(lang, visn)) + lang):  d_lang = residual + query side, d_visn <- key / value side
            d_lang, d_visn_kv = bt.attention_bwd(va.att, t_al, d_ctx_l, True, d_hidden_res=d_lang_res)
            # visn1 = LN(dense(att(visn, lang)) + visn):  d_visn = residual + query side (+ the kv side above), d_lang += kv side
            d_visn, d_lang = bt.attention_bwd(vc.att, t_av, d_ctx_v, True, d_hidden_res=d_visn_res + d_visn_kv, d_ctx_res=d_lang)
            return d_lang, d_visn
        main = torch.cuda.current_stream()
        d_lang_q, d_visn_kv = bt.attention_bwd(va.att, t_al, d_ctx_l, True, d_hidden_res=d_lang_res)
        with torch.cuda.stream(side):
            d_visn1 = bt.self_block_bwd(self.visn_self_att, t_sv, bt.ffn_bwd(self.visn_inter, self.visn_output, t_fv, d_visn3))
            d_ctx_v, d_visn_res = bt.dense_add_norm_bwd(vc.output, t_ov, d_visn1)
            d_visn_q, d_lang_kv = bt.attention_bwd(vc.att, t_av, d_ctx_v, True, d_hidden_res=d_visn_res)
        main.wait_stream(side)
        side.wait_stream(main)
        d_lang_kv.record_stream(main)                                        # made on one stream, consumed on the other
        d_visn_kv.record_stream(side)
        d_lang = d_lang_q.add_(d_lang_kv)
        with torch.cuda.stream(side):
            d_visn = d_visn_q.add_(d_visn_kv)
        return d_lang, d_visn


class LxmertVisualFeatureEncoder(nn.Module):                           # lxmert_lrp.py:742-767
    def __init__(self, c):
        super().__init__()
        self.visn_fc = nn.Linear(c.visual_feat_dim, c.hidden_size)
        self.visn_layer_norm = nn.LayerNorm(c.hidden_size, eps=1e-12)
        self.box_fc = nn.Linear(c.visual_pos_dim, c.hidden_size)
        self.box_layer_norm = nn.LayerNorm(c.hidden_size, eps=1e-12)

    def forward(self, visual_feats, visual_pos):
        return (self.visn_layer_norm(self.visn_fc(visual_feats)) + self.box_layer_norm(self.box_fc(visual_pos))) / 2


class LxmertEncoder(nn.Module):                                        # lxmert_lrp.py:774-866
    def __init__(self, c):
        super().__init__()
        self.visn_fc = LxmertVisualFeatureEncoder(c)
        self.config = c
        self.num_l_layers, self.num_x_layers, self.num_r_layers = c.l_layers, c.x_layers, c.r_layers
        self.layer = nn.ModuleList(LxmertLayer(c) for _ in range(c.l_layers))
        self.x_layers = nn.ModuleList(LxmertXLayer(c) for _ in range(c.x_layers))
        self.r_layers = nn.ModuleList(LxmertLayer(c) for _ in range(c.r_layers))

    def forward(self, lang_feats, lang_attention_mask, visual_feats, visual_pos, visual_attention_mask=None):
        visual_feats = self.visn_fc(visual_feats, visual_pos)
        for blk in self.layer:
            lang_feats = blk(lang_feats, lang_attention_mask)[0]
        for blk in self.r_layers:
            visual_feats = blk(visual_feats, visual_attention_mask)[0]
        for blk in self.x_layers:
            lang_feats, visual_feats = blk(lang_feats, lang_attention_mask, visual_feats, visual_attention_mask)[:2]
        return lang_feats, visual_feats

    def relprop(self, cam, **kwargs):
        """lxmert_lrp.py:855-866: x-layers, then the image stream's r-layers, then the language layers (the relevance stops at
        the encoder inputs: the embeddings' rules are not part of the reference's pass)."""
        cam_lang, cam_vis = cam
        for blk in reversed(self.x_layers):
            cam_lang, cam_vis = blk.relprop((cam_lang, cam_vis), **kwargs)
        for blk in reversed(self.r_layers):
            cam_vis = blk.relprop(cam_vis, **kwargs)
        for blk in reversed(self.layer):
            cam_lang = blk.relprop(cam_lang, **kwargs)
        return cam_lang, cam_vis

    # ---- tape path of the explainability pass (bert_tape.py): no autograd graph, no weight gradients
    overlap_modalities = True     # tape path: the image chain of every layer group on a side stream beside the text chain

    def forward_tape(self, lang, lang_mask, visual_feats, visual_pos, visn_mask=None, lang_repeat=1, visn_repeat=1, lang_encoded=False):
        """``lang_repeat`` / ``visn_repeat`` (forward-only callers): the single-modality layers run on the batch as given and
        their output (and mask) is repeated that many times, sample-major, before the cross-modality layers -- the perturbation
        evaluator re-runs a sample 9 times with only ONE modality changed, so the other modality's own layers (9 language / 5
        object-relationship layers: 20 % / 28 % of a forward) are the same for all 9 and run once.

        The 9 language layers and the 5 object-relationship layers are independent of each other, and so are the two chains
        of every cross-modality layer after its cross-attentions (``LxmertXLayer.forward_tape``): at the batch sizes of an
        explainability pass these are 448- / 1152-row GEMMs that each fill a fraction of the chip, so the image chain runs on a
        side stream beside the text chain (fork / join inside one hipGraph when captured)."""
        main = torch.cuda.current_stream()
        side = ops.side_stream(lang.device) if self.overlap_modalities else None
        tapes = {"l": [], "r": [], "x": [], "side": side}
        if side is not None:
            side.wait_stream(main)
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            visn = self.visn_fc(visual_feats, visual_pos)
            for blk in self.r_layers:
                visn, t = bt.layer_fwd(blk, visn, visn_mask)
                tapes["r"].append(t)
        if not lang_encoded:      # (``lang_encoded``: ``lang`` already IS the output of the language layers -- ``encode_language``)
            for blk in self.layer:
                lang, t = bt.layer_fwd(blk, lang, lang_mask)
                tapes["l"].append(t)
        if lang_repeat > 1:
            lang = lang.repeat_interleave(lang_repeat, dim=0)
            lang_mask = None if lang_mask is None else lang_mask.repeat_interleave(lang_repeat, dim=0)
        if visn_repeat > 1:
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                visn = visn.repeat_interleave(visn_repeat, dim=0)
                visn_mask = None if visn_mask is None else visn_mask.repeat_interleave(visn_repeat, dim=0)
        for blk in self.x_layers:
            if side is not None:                                      # each chain reads the other's output of the layer below
                main.wait_stream(side)
                side.wait_stream(main)
            lang, visn, t = blk.forward_tape(lang, lang_mask, visn, visn_mask, side)
            tapes["x"].append(t)
        if side is not None:
            main.wait_stream(side)
        return lang, visn, tapes

    def backward_tape(self, tapes, d_lang, d_visn=None):
        """``d_lang [B, T, E]`` (and optionally ``d_visn``): gradients w.r.t. the encoder outputs; fills the gradient slab of
        every attention block (the lowest block of each stream skips its input gradients: nothing below reads them)."""
        side = tapes.get("side")
        main = torch.cuda.current_stream()
        if side is not None:
            side.wait_stream(main)
        for blk, t in zip(reversed(self.x_layers), reversed(tapes["x"])):
            d_lang, d_visn = blk.backward_tape(t, d_lang, d_visn, side)
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            for i in range(len(self.r_layers) - 1, -1, -1):
                d_visn = bt.layer_bwd(self.r_layers[i], tapes["r"][i], d_visn, need_input=i > 0)
        for i in range(len(self.layer) - 1, -1, -1):
            d_lang = bt.layer_bwd(self.layer[i], tapes["l"][i], d_lang, need_input=i > 0)
        if side is not None:
            main.wait_stream(side)


class LxmertPooler(nn.Module):                                         # lxmert_lrp.py:868-884
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.hidden_size)

    def forward(self, hidden_states):
        self._lrp_tape = None                                       # a stale tape must not outlive this forward
        if torch.is_grad_enabled():
            self._lrp_tape = hidden_states.detach()
        return torch.tanh(self.dense(hidden_states[:, 0]))

    def relprop(self, cam, **kwargs):
        return bert_lrp.pooler_relprop(self, cam)


class LxmertVisualAnswerHead(nn.Module):                               # lxmert_lrp.py:941-953
    def __init__(self, c, num_labels):
        super().__init__()
        h = c.hidden_size
        self.logit_fc = nn.Sequential(nn.Linear(h, h * 2), nn.GELU(), nn.LayerNorm(h * 2, eps=1e-12),
                                      nn.Linear(h * 2, num_labels))

    def forward(self, hidden_states):
        fc = self.logit_fc
        normed = fc[2](fc[1](fc[0](hidden_states)))
        self._lrp_tape = None                                       # a stale tape must not outlive this forward
        if torch.is_grad_enabled():
            self._lrp_tape = (hidden_states.detach(), normed.detach())
        return fc[3](normed)

    def relprop(self, cam, **kwargs):
        """lxmert_lrp.py:955-958: the two Linear rules (GELU and LayerNorm pass relevance through)."""
        x, normed = bert_lrp.tape_of(self)
        cam = lrp.linear_relprop(cam, normed, self.logit_fc[3].weight, normalize=False)
        return lrp.linear_relprop(cam, x, self.logit_fc[0].weight, normalize=False)


def _extended_mask(mask, dtype):
    """``[B, N]`` 1/0 mask -> additive ``[B, 1, 1, N]`` (0 / -10000), lxmert_lrp.py:1188-1207."""
    return None if mask is None else (1.0 - mask[:, None, None, :].to(dtype)) * -10000.0


class LxmertModel(nn.Module):                                          # lxmert_lrp.py:1122-1262
    def __init__(self, c):
        super().__init__()
        self.config = c
        self.embeddings = LxmertEmbeddings(c)
        self.encoder = LxmertEncoder(c)
        self.pooler = LxmertPooler(c)

    def forward(self, input_ids=None, visual_feats=None, visual_pos=None, attention_mask=None,
                visual_attention_mask=None, token_type_ids=None, inputs_embeds=None):
        if visual_feats is None or visual_pos is None:
            raise ValueError("`visual_feats` and `visual_pos` cannot be None")
        if (input_ids is None) == (inputs_embeds is None):
            raise ValueError("specify exactly one of input_ids and inputs_embeds")
        emb = self.embeddings(input_ids, token_type_ids, inputs_embeds)
        if attention_mask is None:
            attention_mask = torch.ones(emb.shape[:2], device=emb.device)
        lang, visn = self.encoder(emb, _extended_mask(attention_mask, emb.dtype), visual_feats, visual_pos,
                                  _extended_mask(visual_attention_mask, emb.dtype))
        return types.SimpleNamespace(language_output=lang, vision_output=visn, pooled_output=self.pooler(lang))

    def relprop(self, cam, **kwargs):
        """lxmert_lrp.py:1253-1257."""
        cam_lang, cam_vis = cam
        return self.encoder.relprop((self.pooler.relprop(cam_lang, **kwargs), cam_vis), **kwargs)


class LxmertForQuestionAnswering(nn.Module):                           # lxmert_lrp.py:1532-1692
    def __init__(self, c):
        super().__init__()
        self.config = c
        self.num_qa_labels = c.num_qa_labels
        self.lxmert = LxmertModel(c)
        self.answer_head = LxmertVisualAnswerHead(c, c.num_qa_labels)

    @property
    def device(self):
        return next(self.parameters()).device

    def forward(self, input_ids=None, visual_feats=None, visual_pos=None, attention_mask=None,
                visual_attention_mask=None, token_type_ids=None, inputs_embeds=None, **unused):
        out = self.lxmert(input_ids, visual_feats, visual_pos, attention_mask, visual_attention_mask, token_type_ids,
                          inputs_embeds)
        out.question_answering_score = self.answer_head(out.pooled_output)
        self.vis_shape = out.vision_output.shape                       # lxmert_lrp.py:1677
        return out

    def relprop(self, cam, **kwargs):
        """``model.relprop(one_hot, alpha=1)`` (lxmert_lrp.py:1689-1692): the LRP pass of the forward that just ran with grad
        mode on; fills ``get_attn_cam()`` of every attention module and returns ``(cam_lang [B, T, E], cam_vis [B, I, E])``.
        Closed-form rules (``lrp.py`` / ``bert_lrp.py``), the attention cores on the HIP kernels of ``csrc/attention_lrp.hip``."""
        if kwargs.get("alpha", 1) != 1:
            raise NotImplementedError("the generators call relprop with alpha = 1 (lxmert/.../ExplanationGenerator.py:137)")
        with torch.no_grad():
            cam_lang = self.answer_head.relprop(cam.to(torch.float32), **kwargs)
            cam_vis = torch.zeros(self.vis_shape, dtype=torch.float32, device=cam_lang.device)
            return self.lxmert.relprop((cam_lang, cam_vis), **kwargs)

    # ---- tape path of the explainability pass: same scores, every attention block's P in its slab; ``backward_tape`` fills
    # the gradient slabs from d(scores) without an autograd graph through the encoder (bert_tape.py)
    def forward_tape(self, input_ids=None, visual_feats=None, visual_pos=None, attention_mask=None,
                     visual_attention_mask=None, token_type_ids=None, inputs_embeds=None, **unused):
        m = self.lxmert
        with torch.no_grad():
            emb = m.embeddings(input_ids, token_type_ids, inputs_embeds)
            if attention_mask is None:
                attention_mask = torch.ones(emb.shape[:2], device=emb.device)
            lang, visn, tapes = m.encoder.forward_tape(emb, _extended_mask(attention_mask, emb.dtype), visual_feats, visual_pos,
                                                       _extended_mask(visual_attention_mask, emb.dtype))
        # the head (pooler + answer head on the [CLS] row, B x 768) goes through autograd: a handful of tiny ops
        cls = lang[:, 0].detach().requires_grad_(True)
        with torch.enable_grad():
            scores = self.answer_head(torch.tanh(m.pooler.dense(cls)))
        return scores, (tapes, cls, scores, lang.shape)

    @torch.no_grad()
    def encode_language(self, input_ids, attention_mask=None, token_type_ids=None):
        """Output of the embeddings + the 9 language-only layers, grad-free (tape forward): what ``scores_no_grad(lang_encoded=...)``
        takes when SEVERAL region sets are scored against the same questions (the perturbation evaluator's step groups)."""
        m = self.lxmert
        emb = m.embeddings(input_ids, token_type_ids, None)
        if attention_mask is None:
            attention_mask = torch.ones(emb.shape[:2], device=emb.device)
        lang_mask = _extended_mask(attention_mask, emb.dtype)
        for blk in m.encoder.layer:
            emb, _ = bt.layer_fwd(blk, emb, lang_mask)
        return emb

    @torch.no_grad()
    def scores_no_grad(self, input_ids=None, visual_feats=None, visual_pos=None, attention_mask=None,
                       visual_attention_mask=None, token_type_ids=None, inputs_embeds=None, lang_repeat=1, visn_repeat=1,
                       lang_encoded=None, **unused):
        """``forward(...).question_answering_score`` through the tape forward (packed q / k / v GEMMs, fused bias / add /
        LayerNorm, the two modalities side by side) without keeping anything for a backward -- what the perturbation evaluator's
        re-runs need (lxmert/lxmert/perturbation.py:119-131).  An empty region set takes the module forward.
        ``lang_repeat`` / ``visn_repeat``: the text (visual) inputs are given ONCE per sample and their own layers' output is
        repeated for the cross-modality layers (``LxmertEncoder.forward_tape``); the other modality comes already repeated.
        ``lang_encoded``: the output of ``encode_language`` for these questions -- the embeddings and language layers are skipped."""
        if visual_feats.shape[1] == 0 or (input_ids is None and lang_encoded is None):
            if lang_repeat != 1 or visn_repeat != 1 or lang_encoded is not None:
                raise ValueError("scores_no_grad: the repeat hints need a non-empty region set and token ids")
            return self.forward(input_ids, visual_feats, visual_pos, attention_mask, visual_attention_mask, token_type_ids,
                                inputs_embeds).question_answering_score
        m = self.lxmert
        emb = lang_encoded if lang_encoded is not None else m.embeddings(input_ids, token_type_ids, inputs_embeds)
        if attention_mask is None:
            attention_mask = torch.ones(emb.shape[:2], device=emb.device)
        lang, _, _ = m.encoder.forward_tape(emb, _extended_mask(attention_mask, emb.dtype), visual_feats, visual_pos,
                                            _extended_mask(visual_attention_mask, emb.dtype), lang_repeat, visn_repeat,
                                            lang_encoded=lang_encoded is not None)
        return self.answer_head(m.pooler(lang))

    @torch.no_grad()
    def backward_tape(self, state, d_scores):
        """``d_scores [B, answers]`` (one-hot seeds): d(sum seeds * scores)/dP into every attention module's gradient slab."""
        tapes, cls, scores, lang_shape = state
        with torch.enable_grad():
            (d_cls,) = torch.autograd.grad(scores, cls, d_scores, retain_graph=True)
        d_lang = torch.zeros(lang_shape, dtype=torch.float32, device=d_cls.device)
        d_lang[:, 0] = d_cls                                           # only the [CLS] row feeds the pooler
        self.lxmert.encoder.backward_tape(tapes, d_lang, None)
