"""LXMERT relevancy generators -- the reference's ``lxmert/lxmert/src/ExplanationGenerator.py`` surface on the HIP kernels.

``GeneratorOurs(model_usage).generate_ours(input, index=None, use_lrp=True, normalize_self_attention=True,
apply_self_in_rule_10=True, method_name="ours")`` keeps the reference signature and duck-typed ``model_usage``
(``.forward(item) -> obj.question_answering_score``, ``.model``, ``.text_len``, ``.image_boxes_len``;
lxmert/lxmert/perturbation.py:45-83).  Attention modules are reached through the same attribute paths
(``model.lxmert.encoder.{layer, r_layers, x_layers}`` ...) and must expose ``get_attn()`` / ``get_attn_gradients()``
``[1, H, Nq, Nk]`` device tensors (``attention_modules.BertStyleAttention`` serves them from the capture slabs).

Underneath: the 9 language and 5 vision self-attention layers are one chain launch each (rules 6+7 carry the
cross matrices as second right-hand side), the cross layers use the rule-10/11 kernels; both cross directions are
computed from the pre-update state and added afterwards, exactly like the reference (:176-196).
LRP methods (``use_lrp=True`` -- the default argument --, ``generate_transformer_attr``, ``generate_partial_lrp``) run the
reference's schedule on ``get_attn_cam()``, filled by the body's LRP pass ``model.relprop(one_hot, alpha=1)``
(lxmert/lxmert/src/lxmert_lrp.py:422-461, 1689-1692).  ``lxmert_model.LxmertForQuestionAnswering.relprop`` is that pass
(``bert_lrp.py``: closed-form rules around the HIP attention-core kernels); a body without ``relprop`` raises
``NotImplementedError`` naming the missing method before any work is done.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops, rules
from .rules import avg_heads, compute_rollout_attention, handle_residual  # noqa: F401

apply_self_attention_rules = rules.apply_self_attention_rules
apply_mm_attention_rules = rules.apply_mm_attention_rules_lxmert


def _pair(module, use_lrp=False):
    """``(cam, grad)``: the LRP cam on the LRP route, the attention probabilities otherwise (reference :64-67 etc.)."""
    return (module.get_attn_cam() if use_lrp else module.get_attn()).detach(), module.get_attn_gradients().detach()


def _require_relprop(model):
    if not hasattr(model, "relprop"):
        raise NotImplementedError(
            "use_lrp=True / transformer_attr / partial_lrp read LRP attention cams (get_attn_cam) that the body's "
            "relprop() must produce; %s has no relprop().  Use use_lrp=False (the evaluator's 'ours_no_lrp'), or plug a "
            "body built on an LRP layer library (reference: lxmert/lxmert/src/lxmert_lrp.py)." % type(model).__name__)


def _backward_on_answer(model_usage, input, index, use_lrp=False, backward=True):
    """lxmert/.../ExplanationGenerator.py:136,153-165: forward, one-hot on the answer, backward [, the body's relprop]."""
    model = model_usage.model
    if use_lrp:
        _require_relprop(model)
    output = rules.forward_for_backward(model, lambda: model_usage.forward(input).question_answering_score)
    if index is None:
        index = np.argmax(output.cpu().data.numpy(), axis=-1)
    one_hot = torch.zeros_like(output)
    one_hot[0, index] = 1
    if backward:
        loss = torch.sum(one_hot * output)
        model.zero_grad()
        loss.backward(retain_graph=True)
    if use_lrp:
        model.relprop(one_hot.detach().clone(), alpha=1)
    return model


class GeneratorOurs:
    def __init__(self, model_usage, save_visualization=False):
        """``save_visualization`` is STORED and never read, exactly as in the reference (lxmert/lxmert/src/ExplanationGenerator.py:57-59,
        :216-218, :369-371 assign it; no method of that file tests it or writes an image) -- the images of the LXMERT notebook are drawn
        by the notebook itself from the returned ``R_t_i``.  (VisualBERT's generators DO act on their flag: ``visualbert_explainability``.)"""
        self.model_usage = model_usage
        self.save_visualization = save_visualization
        self.fused = True   # one-launch schedule kernel when T, I <= 48; False forces the per-rule kernels
        self.use_tape = True   # generate_ours_batch: hand-written forward / backward of the body when it offers one

    def _generate_ours_fused(self, model):
        """All 38 rule applications in ONE kernel launch (``mmx_lxmert_schedule``); same results as the per-rule path."""
        enc = model.lxmert.encoder
        xs = list(enc.x_layers)
        u = self.use_lrp
        R_tt, R_ti, R_ii, R_it = ops.lxmert_schedule(
            [_pair(b.attention.self, u) for b in enc.layer], [_pair(b.attention.self, u) for b in enc.r_layers],
            [_pair(b.visual_attention.att, u) for b in xs], [_pair(b.visual_attention_copy.att, u) for b in xs[:-1]],
            [_pair(b.lang_self_att.self, u) for b in xs], [_pair(b.visn_self_att.self, u) for b in xs[:-1]],
            apply_normalization=self.normalize_self_attention, apply_self_in_rule_10=self.apply_self_in_rule_10)
        self.R_t_t, self.R_t_i, self.R_i_i, self.R_i_t = R_tt[0], R_ti[0], R_ii[0], R_it[0]
        return self.R_t_t, self.R_t_i

    def generate_ours_batch(self, model_inputs, index=None, normalize_self_attention=True, apply_self_in_rule_10=True,
                            check_diag=True):
        """B samples in ONE forward + ONE backward (B one-hot seeds) + ONE schedule launch.

        The evaluator (``perturbation.py:216-250``) explains one item per call; every per-item pass is ~1000 launches on
        a batch-1 body, i.e. bound by the host, not by the GPU.  Samples are independent (sample b's score depends on
        sample b's inputs only), so a batch of B one-hot seeds in one backward leaves exactly the per-sample
        gradients in the slabs, and the schedule kernel runs one workgroup per sample.

        Question lengths may DIFFER inside the batch: pad ``input_ids`` / ``token_type_ids`` to a common ``T`` and mark
        the real tokens in ``attention_mask`` (1 = real, left-aligned).  Padded keys carry zero probability and padded
        query rows zero gradient; the schedule kernel runs sample b's rules on its leading ``attention_mask[b].sum()``
        tokens only (a padded row inside ``handle_residual`` would be the 0/0 the reference's own assert guards
        against) and returns zeros beyond them.

        ``model_inputs``: the keyword tensors of the model, batch-first (``input_ids [B, T]``, ``visual_feats [B, I, F]``,
        ...).  ``index``: ``None`` (arg-max answer per sample, chosen on the device) or ``[B]`` answer ids.
        ``check_diag``: ``True`` asserts ``handle_residual``'s ``diag >= 0`` now (one device->host read); ``"defer"``
        leaves the device word in ``self.diag_min`` and synchronises nothing (``GraphedGenerateOursBatch`` uses it).
        Returns ``(R_t_t [B, T, T], R_t_i [B, T, I])``; ``self.R_i_i`` / ``self.R_i_t`` hold the image-side matrices.
        """
        self.use_lrp = False
        self.normalize_self_attention = normalize_self_attention
        self.apply_self_in_rule_10 = apply_self_in_rule_10
        model = self.model_usage.model
        if self.use_tape and hasattr(model, "forward_tape"):
            # tape path (lxmert_model.forward_tape / backward_tape, bert_tape.py): no autograd graph through the encoder, packed
            # q/k/v GEMMs, fused add + LayerNorm, residual adds folded into GEMMs -- about half the launches of the route below
            output, state = model.forward_tape(**model_inputs)                                           # [B, A]
            idx = output.argmax(dim=-1) if index is None else torch.as_tensor(index, device=output.device).reshape(-1)
            one_hot = torch.zeros_like(output).scatter_(1, idx.reshape(-1, 1), 1.0)
            model.backward_tape(state, one_hot)
        else:
            output = rules.forward_for_backward(model, lambda: model(**model_inputs).question_answering_score)   # [B, A]
            idx = output.argmax(dim=-1) if index is None else torch.as_tensor(index, device=output.device).reshape(-1)
            one_hot = torch.zeros_like(output).scatter_(1, idx.reshape(-1, 1), 1.0)
            model.zero_grad()
            torch.sum(one_hot * output).backward(retain_graph=True)
        T, I = model_inputs["input_ids"].shape[1], model_inputs["visual_feats"].shape[1]
        if max(T, I) > ops.LXMERT_FUSED_MAX_TOKENS:
            raise NotImplementedError("generate_ours_batch runs the one-launch schedule (T, I <= %d)"
                                      % ops.LXMERT_FUSED_MAX_TOKENS)
        mask = model_inputs.get("attention_mask")
        text_len = mask.sum(dim=1).to(torch.int32) if mask is not None else None       # stays on the device
        enc = model.lxmert.encoder
        xs = list(enc.x_layers)
        out = ops.lxmert_schedule(
            [_pair(b.attention.self) for b in enc.layer], [_pair(b.attention.self) for b in enc.r_layers],
            [_pair(b.visual_attention.att) for b in xs], [_pair(b.visual_attention_copy.att) for b in xs[:-1]],
            [_pair(b.lang_self_att.self) for b in xs], [_pair(b.visn_self_att.self) for b in xs[:-1]],
            apply_normalization=normalize_self_attention, apply_self_in_rule_10=apply_self_in_rule_10,
            check_diag=check_diag, text_len=text_len)
        self.R_t_t, self.R_t_i, self.R_i_i, self.R_i_t = out[:4]
        self.diag_min = out[4] if check_diag == "defer" else None
        return self.R_t_t, self.R_t_i

    # ---- single-stream pieces: rules 6+7 for a list of blocks in one chain launch
    def _self_chain(self, pairs, R_ss, R_sq):
        attn = [a for a, _ in pairs]
        grad = [g for _, g in pairs]
        R_ss, R_sq = ops.relevancy_self_chain(attn, grad, 1, R_init=R_ss, R_sq_init=R_sq)
        return R_ss[0], R_sq[0]

    def handle_self_attention_lang(self, blocks):
        self.R_t_t, self.R_t_i = self._self_chain([_pair(b.attention.self, self.use_lrp) for b in blocks], self.R_t_t, self.R_t_i)

    def handle_self_attention_image(self, blocks):
        self.R_i_i, self.R_i_t = self._self_chain([_pair(b.attention.self, self.use_lrp) for b in blocks], self.R_i_i, self.R_i_t)

    def handle_co_attn_self_lang(self, block):
        self.R_t_t, self.R_t_i = self._self_chain([_pair(block.lang_self_att.self, self.use_lrp)], self.R_t_t, self.R_t_i)

    def handle_co_attn_self_image(self, block):
        self.R_i_i, self.R_i_t = self._self_chain([_pair(block.visn_self_att.self, self.use_lrp)], self.R_i_i, self.R_i_t)

    def handle_co_attn_lang(self, block):
        cam_t_i = avg_heads(*_pair(block.visual_attention.att, self.use_lrp))
        return apply_mm_attention_rules(self.R_t_t, self.R_i_i, self.R_i_t, cam_t_i,
                                        apply_normalization=self.normalize_self_attention,
                                        apply_self_in_rule_10=self.apply_self_in_rule_10)

    def handle_co_attn_image(self, block):
        cam_i_t = avg_heads(*_pair(block.visual_attention_copy.att, self.use_lrp))
        return apply_mm_attention_rules(self.R_i_i, self.R_t_t, self.R_t_i, cam_i_t,
                                        apply_normalization=self.normalize_self_attention,
                                        apply_self_in_rule_10=self.apply_self_in_rule_10)

    def generate_ours(self, input, index=None, use_lrp=True, normalize_self_attention=True, apply_self_in_rule_10=True,
                      method_name="ours"):
        self.use_lrp = use_lrp
        self.normalize_self_attention = normalize_self_attention
        self.apply_self_in_rule_10 = apply_self_in_rule_10
        model = _backward_on_answer(self.model_usage, input, index, use_lrp)
        text_tokens = self.model_usage.text_len
        image_bboxes = self.model_usage.image_boxes_len
        dev = model.device
        if max(text_tokens, image_bboxes) <= ops.LXMERT_FUSED_MAX_TOKENS and self.fused:
            return self._generate_ours_fused(model)
        self.R_t_t = torch.eye(text_tokens, text_tokens, device=dev)
        self.R_i_i = torch.eye(image_bboxes, image_bboxes, device=dev)
        self.R_t_i = torch.zeros(text_tokens, image_bboxes, device=dev)
        self.R_i_t = torch.zeros(image_bboxes, text_tokens, device=dev)

        self.handle_self_attention_lang(model.lxmert.encoder.layer)
        self.handle_self_attention_image(model.lxmert.encoder.r_layers)
        blocks = model.lxmert.encoder.x_layers
        for i, blk in enumerate(blocks):
            if i == len(blocks) - 1:
                break
            R_t_i_addition, R_t_t_addition = self.handle_co_attn_lang(blk)
            R_i_t_addition, R_i_i_addition = self.handle_co_attn_image(blk)
            self.R_t_i = self.R_t_i + R_t_i_addition
            self.R_t_t = self.R_t_t + R_t_t_addition
            self.R_i_t = self.R_i_t + R_i_t_addition
            self.R_i_i = self.R_i_i + R_i_i_addition
            self.handle_co_attn_self_lang(blk)
            self.handle_co_attn_self_image(blk)
        blk = blocks[-1]
        R_t_i_addition, R_t_t_addition = self.handle_co_attn_lang(blk)
        self.R_t_i = self.R_t_i + R_t_i_addition
        self.R_t_t = self.R_t_t + R_t_t_addition
        self.handle_co_attn_self_lang(blk)
        self.R_t_t[0, 0] = 0   # disregard the [CLS] token itself (:210)
        return self.R_t_t, self.R_t_i


class GeneratorOursAblationNoAggregation:
    """Reference :215-365: every accumulation replaced by assignment."""

    def __init__(self, model_usage, save_visualization=False):
        self.model_usage = model_usage
        self.save_visualization = save_visualization

    def _self(self, module, R_ss, R_sq):
        cam = avg_heads(*_pair(module, self.use_lrp))
        return apply_self_attention_rules(R_ss, R_sq, cam)

    def generate_ours_no_agg(self, input, index=None, use_lrp=False, normalize_self_attention=True,
                             method_name="ours_no_agg"):
        self.use_lrp = use_lrp
        self.normalize_self_attention = normalize_self_attention
        model = _backward_on_answer(self.model_usage, input, index, use_lrp)
        T, I = self.model_usage.text_len, self.model_usage.image_boxes_len
        dev = model.device
        self.R_t_t, self.R_i_i = torch.eye(T, T, device=dev), torch.eye(I, I, device=dev)
        self.R_t_i, self.R_i_t = torch.zeros(T, I, device=dev), torch.zeros(I, T, device=dev)
        for blk in model.lxmert.encoder.layer:
            self.R_t_t, self.R_t_i = self._self(blk.attention.self, self.R_t_t, self.R_t_i)
        for blk in model.lxmert.encoder.r_layers:
            self.R_i_i, self.R_i_t = self._self(blk.attention.self, self.R_i_i, self.R_i_t)

        def co_lang(blk):
            cam = avg_heads(*_pair(blk.visual_attention.att, self.use_lrp))
            return apply_mm_attention_rules(self.R_t_t, self.R_i_i, self.R_i_t, cam,
                                            apply_normalization=self.normalize_self_attention)

        def co_img(blk):
            cam = avg_heads(*_pair(blk.visual_attention_copy.att, self.use_lrp))
            return apply_mm_attention_rules(self.R_i_i, self.R_t_t, self.R_t_i, cam,
                                            apply_normalization=self.normalize_self_attention)

        blocks = model.lxmert.encoder.x_layers
        for i, blk in enumerate(blocks):
            if i == len(blocks) - 1:
                break
            t_i, t_t = co_lang(blk)
            i_t, i_i = co_img(blk)
            self.R_t_i, self.R_t_t, self.R_i_t, self.R_i_i = t_i, t_t, i_t, i_i
            self.R_t_t, self.R_t_i = self._self(blk.lang_self_att.self, self.R_t_t, self.R_t_i)
            self.R_i_i, self.R_i_t = self._self(blk.visn_self_att.self, self.R_i_i, self.R_i_t)
        blk = blocks[-1]
        self.R_t_i, self.R_t_t = co_lang(blk)
        self.R_t_t, self.R_t_i = self._self(blk.lang_self_att.self, self.R_t_t, self.R_t_i)
        self.R_t_t[0, 0] = 0
        return self.R_t_t, self.R_t_i


class GeneratorBaselines:
    """Attention-only baselines of the reference (:368-665): raw attention, attention GradCAM, rollout.
    ``generate_transformer_attr`` / ``generate_partial_lrp`` read LRP cams: they need a body with ``relprop`` (module docstring)."""

    def __init__(self, model_usage, save_visualization=False):
        self.model_usage = model_usage
        self.save_visualization = save_visualization

    @staticmethod
    def _head_mean(module):
        cam = module.get_attn().detach()
        return cam.reshape(-1, cam.shape[-2], cam.shape[-1]).mean(dim=0)

    def generate_raw_attn(self, input, method_name="raw_attention"):
        """Reference :508-540: head-means of the last cross-attention (``R_t_i``) and last language self-attention."""
        self.model_usage.forward(input)
        model = self.model_usage.model
        blk = model.lxmert.encoder.x_layers[-1]
        self.R_t_i = self._head_mean(blk.visual_attention.att)
        self.R_t_t = self._head_mean(blk.lang_self_att.self)
        self.R_t_t[0, 0] = 0
        return self.R_t_t, self.R_t_i

    def gradcam(self, cam, grad):
        return rules.gradcam(cam, grad)

    def generate_attn_gradcam(self, input, index=None, method_name="gradcam"):
        """Reference :549-592: GradCAM of the last cross-attention and the last language self-attention."""
        model = _backward_on_answer(self.model_usage, input, index)
        blk = model.lxmert.encoder.x_layers[-1]
        att = blk.visual_attention.att
        self.R_t_i = self.gradcam(att.get_attn().detach(), att.get_attn_gradients().detach())
        sa = blk.lang_self_att.self
        self.R_t_t = self.gradcam(sa.get_attn().detach(), sa.get_attn_gradients().detach())
        self.R_t_t[0, 0] = 0
        return self.R_t_t, self.R_t_i

    def generate_rollout(self, input, method_name="rollout"):
        """Reference :594-665.  ``R_t_i = R_t_t^T . (cam_t_i . R_i_i)`` uses the text rollout WITHOUT the last language
        self-attention block; the returned ``R_t_t`` includes it."""
        self.model_usage.forward(input)
        model = self.model_usage.model
        enc = model.lxmert.encoder
        cams_text = [self._head_mean(b.attention.self) for b in enc.layer]
        cams_image = [self._head_mean(b.attention.self) for b in enc.r_layers]
        for i, blk in enumerate(enc.x_layers):
            if i == len(enc.x_layers) - 1:
                break
            cams_text.append(self._head_mean(blk.lang_self_att.self))
            cams_image.append(self._head_mean(blk.visn_self_att.self))
        blk = enc.x_layers[-1]
        cam_t_i = self._head_mean(blk.visual_attention.att)
        self.R_t_t = compute_rollout_attention(cams_text)
        self.R_i_i = compute_rollout_attention(cams_image)
        self.R_t_i = ops.matmul(self.R_t_t, ops.matmul(cam_t_i, self.R_i_i), trans_a=True)
        cams_text.append(self._head_mean(blk.lang_self_att.self))
        self.R_t_t = compute_rollout_attention(cams_text)
        self.R_t_t[0, 0] = 0
        return self.R_t_t, self.R_t_i

    def generate_transformer_attr(self, input, index=None, method_name="transformer_attr"):
        """Reference :373-460: rule 6 only (``R_ss += cam.R_ss``) per stream on the LRP cams -- one chain launch per
        stream -- and ``R_t_i`` = rule 5 of the last cross-attention's cam."""
        model = _backward_on_answer(self.model_usage, input, index, use_lrp=True)
        enc = model.lxmert.encoder
        xs = list(enc.x_layers)
        lang = [_pair(b.attention.self, True) for b in enc.layer] + [_pair(b.lang_self_att.self, True) for b in xs]
        img = [_pair(b.attention.self, True) for b in enc.r_layers] + [_pair(b.visn_self_att.self, True) for b in xs[:-1]]
        self.R_t_t = ops.relevancy_self_chain([a for a, _ in lang], [g for _, g in lang], 1)[0]
        self.R_i_i = ops.relevancy_self_chain([a for a, _ in img], [g for _, g in img], 1)[0]
        self.R_t_i = avg_heads(*_pair(xs[-1].visual_attention.att, True))
        self.R_i_t = torch.zeros(self.R_i_i.shape[0], self.R_t_t.shape[0], device=self.R_t_t.device)
        self.R_t_t[0, 0] = 0
        return self.R_t_t, self.R_t_i

    def generate_partial_lrp(self, input, index=None, method_name="partial_lrp"):
        """Reference :462-506: head-means of the last x-layer's LRP cams, each min-max normalised (no backward)."""
        model = _backward_on_answer(self.model_usage, input, index, use_lrp=True, backward=False)
        blk = model.lxmert.encoder.x_layers[-1]

        def mean_cam(module):
            cam = module.get_attn_cam().detach()
            cam = cam.reshape(-1, cam.shape[-2], cam.shape[-1]).mean(dim=0)
            return (cam - cam.min()) / (cam.max() - cam.min())

        self.R_t_i = mean_cam(blk.visual_attention.att)
        self.R_t_t = mean_cam(blk.lang_self_att.self)
        self.R_t_t[0, 0] = 0
        return self.R_t_t, self.R_t_i


class GraphedGenerateOursBatch:
    """``GeneratorOurs.generate_ours_batch`` captured once into a hipGraph and replayed.

    A batched explain pass is ~1000 launches (19 captured attention layers, each a handful of body ops forward and
    backward) and costs the same ~19 ms at B = 8 and at B = 32: it is bound by the host.  Because the schedule kernel takes
    per-sample question lengths, ONE graph captured at a padded length ``T`` serves every batch whose questions are at
    most ``T`` tokens long -- no grouping by length, no re-capture.  Nothing in the captured pass synchronises: the
    reference's ``handle_residual`` assert becomes the device word ``diag_min``, checked by ``__call__`` when it hands out
    the results (``check=True``, one device->host read per batch instead of one per rule; ``check="deferred"``: read behind the
    replay, asserted at the next call or ``finish()`` -- no stall between this graph and whatever the caller launches next).

        run = GraphedGenerateOursBatch(model, example_inputs)          # example: padded [B, T] ids, [B, I, F] features ...
        R_t_t, R_t_i = run(inputs)                                      # same shapes; returns the graph's output buffers
    """

    def __init__(self, model, example_inputs, index=None, normalize_self_attention=True, apply_self_in_rule_10=True,
                 warmup=2):
        self.static = {k: v.clone() for k, v in example_inputs.items()}
        self.static_index = None if index is None else torch.as_tensor(index, device=self.static["input_ids"].device).clone()
        self.gen = GeneratorOurs(type("Usage", (), {"model": model})())
        kw = dict(normalize_self_attention=normalize_self_attention, apply_self_in_rule_10=apply_self_in_rule_10,
                  check_diag="defer")
        self._call = lambda: self.gen.generate_ours_batch(self.static, self.static_index, **kw)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._call()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with ops.graph_capture(self.graph):
            self.outputs = self._call()
            self.diag_min = self.gen.diag_min
            self.R_i_i, self.R_i_t = self.gen.R_i_i, self.gen.R_i_t
        self._pinned = ops.pinned_state(model)     # slabs / scratch the graph has raw addresses of (see ops.pinned_state)

    def __call__(self, inputs=None, index=None, check=True):
        if inputs is not None:
            for k, v in inputs.items():
                if v.shape != self.static[k].shape:
                    raise ValueError("%s: %s, but the graph was captured for %s (pad the batch to the captured shape)"
                                     % (k, tuple(v.shape), tuple(self.static[k].shape)))
                self.static[k].copy_(v)
        if index is not None:
            if self.static_index is None:
                raise ValueError("the graph was captured with index=None (arg-max answers)")
            self.static_index.copy_(torch.as_tensor(index))
        self.graph.replay()
        if check == "deferred":
            # the same assert WITHOUT stalling this thread on the replay it has just launched (an evaluator wants to launch the next
            # graph now): the word is copied to pinned host memory behind the replay and checked at the NEXT call / ``finish()``
            self.finish()
            if self.diag_min is not None:
                if self._diag_host is None:
                    self._diag_host = [torch.empty(1, dtype=self.diag_min.dtype).pin_memory() for _ in range(2)]
                slot = self._diag_host[self._diag_turn]
                self._diag_turn ^= 1
                slot.copy_(self.diag_min.reshape(1), non_blocking=True)
                done = torch.cuda.Event()
                done.record()
                self._diag_pending = (slot, done)
        elif check and self.diag_min is not None:
            assert self.diag_min.item() >= 0        # the reference's handle_residual assert, once per batch
        return self.outputs

    _diag_host, _diag_turn, _diag_pending = None, 0, None

    def finish(self):
        """Check the ``handle_residual`` word of the last ``check="deferred"`` call (a no-op when nothing is pending)."""
        if self._diag_pending is not None:
            slot, done = self._diag_pending
            self._diag_pending = None
            done.synchronize()
            assert float(slot[0]) >= 0
