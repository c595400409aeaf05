"""DETR relevancy generators -- the reference's ``DETR/modules/ExplanationGenerator.py`` surface on the HIP kernels.

Same class / method names, arguments, defaults, return shapes and attribute side effects (``self.R_i_i``,
``self.R_q_q``, ``self.R_q_i``) as the reference.  ``model`` is duck-typed exactly as there:
``model(img) -> {'pred_logits': [1, Q, C+1]}``, ``model.transformer.encoder.layers[i].self_attn``,
``model.transformer.decoder.layers[i].{self_attn, multihead_attn}``, each with ``get_attn()`` /
``get_attn_gradients()`` returning ``[B*H, Nq, Nk]`` device tensors (``attention_modules.MultiheadAttention`` provides
them straight from the capture slabs).

What changes underneath: the 6-layer encoder chain is ONE ``relevancy_self_chain`` call, each decoder self-attention
block is one call carrying both right-hand sides (rules 6+7), rule 10 runs on the MFMA matmul + row-normalise kernels.
LRP variants (``use_lrp=True``, ``generate_transformer_att``, ``generate_partial_lrp``) need the reference's LRP layer
library and are out of scope: they raise NotImplementedError.
"""
from __future__ import annotations

import torch

from . import ops, rules
from .rules import avg_heads, compute_rollout_attention, handle_residual  # noqa: F401  (re-exported like the reference module)

apply_self_attention_rules = rules.apply_self_attention_rules
apply_mm_attention_rules = rules.apply_mm_attention_rules_detr


def _logits_for_backward(model, img):
    """Forward pass the one-hot backward will run through; no rule reads a weight gradient, so parameters are frozen
    while the graph is built whenever the body allows it (``rules.forward_for_backward``)."""
    return rules.forward_for_backward(model, lambda: model(img)["pred_logits"])


def _one_hot_backward(model, outputs, target_index, index):
    """DETR/modules/ExplanationGenerator.py:153-163: one-hot on (query, class), zero_grad, backward."""
    if index is None:
        index = outputs[0, target_index, :-1].max(1)[1]
    one_hot = torch.zeros_like(outputs)
    one_hot[0, target_index, index] = 1
    loss = torch.sum(one_hot * outputs)
    model.zero_grad()
    loss.backward(retain_graph=True)
    return index


class Generator:
    def __init__(self, model):
        self.model = model
        self.model.eval()

    def forward(self, input_ids, attention_mask):
        return self.model(input_ids, attention_mask)

    # ------------------------------------------------------------------ ours (rules 5, 6, 7, 10)
    def handle_self_attention_image(self, blocks):
        """Encoder chain (reference :110-118) in one launch."""
        attn = [blk.self_attn.get_attn().detach() for blk in blocks]
        grad = [blk.self_attn.get_attn_gradients().detach() for blk in blocks]
        self.R_i_i = ops.relevancy_self_chain(attn, grad, 1, R_init=self.R_i_i)[0]

    def handle_co_attn_self_query(self, block):
        """Rules 6+7 (reference :120-129): ``R_q_q += cam@R_q_q; R_q_i += cam@R_q_i``."""
        attn = block.self_attn.get_attn().detach()
        grad = block.self_attn.get_attn_gradients().detach()
        R_qq, R_qi = ops.relevancy_self_chain([attn], [grad], 1, R_init=self.R_q_q, R_sq_init=self.R_q_i)
        self.R_q_q, self.R_q_i = R_qq[0], R_qi[0]

    def handle_co_attn_query(self, block):
        """Rule 10 (reference :131-140)."""
        cam_q_i = avg_heads(block.multihead_attn.get_attn().detach(), block.multihead_attn.get_attn_gradients().detach())
        self.R_q_i = self.R_q_i + apply_mm_attention_rules(self.R_q_q, self.R_i_i, cam_q_i,
                                                           apply_normalization=self.normalize_self_attention,
                                                           apply_self_in_rule_10=self.apply_self_in_rule_10)

    def generate_ours(self, img, target_index, index=None, use_lrp=True, normalize_self_attention=True,
                      apply_self_in_rule_10=True):
        if use_lrp:
            raise NotImplementedError("use_lrp=True needs the reference's LRP layer library (model.relprop); "
                                      "call with use_lrp=False (the evaluator default 'ours_no_lrp')")
        self.use_lrp = use_lrp
        self.normalize_self_attention = normalize_self_attention
        self.apply_self_in_rule_10 = apply_self_in_rule_10
        outputs = _logits_for_backward(self.model, img)
        _one_hot_backward(self.model, outputs, target_index, index)

        decoder_blocks = self.model.transformer.decoder.layers
        encoder_blocks = self.model.transformer.encoder.layers
        ref = encoder_blocks[0].self_attn.get_attn()
        image_bboxes = ref.shape[-1]
        queries_num = decoder_blocks[0].self_attn.get_attn().shape[-1]
        self.R_i_i = torch.eye(image_bboxes, image_bboxes, device=ref.device)
        self.R_q_q = torch.eye(queries_num, queries_num, device=ref.device)
        self.R_q_i = torch.zeros(queries_num, image_bboxes, device=ref.device)

        self.handle_self_attention_image(encoder_blocks)
        for blk in decoder_blocks:
            self.handle_co_attn_self_query(blk)
            self.handle_co_attn_query(blk)
        aggregated = self.R_q_i.unsqueeze_(0)
        return aggregated[:, target_index, :].unsqueeze_(0).detach()

    # ------------------------------------------------------------------ baselines on the same slabs
    def generate_raw_attn(self, img, target_index):
        """Reference :225-238: head-mean of the last decoder cross-attention."""
        self.model(img)
        cam_q_i = self.model.transformer.decoder.layers[-1].multihead_attn.get_attn().detach()
        cam_q_i = cam_q_i.reshape(-1, cam_q_i.shape[-2], cam_q_i.shape[-1]).mean(dim=0)
        self.R_q_i = cam_q_i
        return self.R_q_i.unsqueeze_(0)[:, target_index, :].unsqueeze_(0)

    def generate_rollout(self, img, target_index):
        """Reference :240-273: rollout of the encoder / decoder self-attention, joined through the last cross-attention."""
        self.model(img)
        decoder_blocks = self.model.transformer.decoder.layers
        encoder_blocks = self.model.transformer.encoder.layers
        cams_image = [blk.self_attn.get_attn().detach().mean(dim=0) for blk in encoder_blocks]
        cams_queries = [blk.self_attn.get_attn().detach().mean(dim=0) for blk in decoder_blocks]
        # the reference indexes ``shape[1]`` of ``[N, N]`` maps here, which only works because they are square
        self.R_i_i = compute_rollout_attention(cams_image)
        self.R_q_q = compute_rollout_attention(cams_queries)
        cam_q_i = decoder_blocks[-1].multihead_attn.get_attn().detach()
        cam_q_i = cam_q_i.reshape(-1, cam_q_i.shape[-2], cam_q_i.shape[-1]).mean(dim=0)
        self.R_q_i = ops.matmul(self.R_q_q, ops.matmul(cam_q_i, self.R_i_i), trans_a=True)
        return self.R_q_i.unsqueeze_(0)[:, target_index, :].unsqueeze_(0)

    def gradcam(self, cam, grad):
        return rules.gradcam(cam, grad)

    def generate_attn_gradcam(self, img, target_index, index=None):
        """Reference :282-305."""
        outputs = _logits_for_backward(self.model, img)
        _one_hot_backward(self.model, outputs, target_index, index)
        last = self.model.transformer.decoder.layers[-1].multihead_attn
        self.R_q_i = self.gradcam(last.get_attn().detach(), last.get_attn_gradients().detach())
        return self.R_q_i.unsqueeze_(0)[:, target_index, :].unsqueeze_(0)

    def generate_transformer_att(self, img, target_index, index=None):
        raise NotImplementedError("transformer_att needs model.relprop (LRP layer library): out of scope")

    def generate_partial_lrp(self, img, target_index, index=None):
        raise NotImplementedError("partial_lrp needs model.relprop (LRP layer library): out of scope")


class GeneratorAlbationNoAgg:
    """No-aggregation ablation (reference :310-403): every ``+=`` of ``generate_ours`` becomes ``=``."""

    def __init__(self, model):
        self.model = model
        self.model.eval()

    def forward(self, input_ids, attention_mask):
        return self.model(input_ids, attention_mask)

    def handle_self_attention_image(self, blocks):
        for blk in blocks:
            cam = avg_heads(blk.self_attn.get_attn().detach(), blk.self_attn.get_attn_gradients().detach())
            self.R_i_i = ops.matmul(cam, self.R_i_i)

    def handle_co_attn_self_query(self, block):
        cam = avg_heads(block.self_attn.get_attn().detach(), block.self_attn.get_attn_gradients().detach())
        self.R_q_q, self.R_q_i = apply_self_attention_rules(self.R_q_q, self.R_q_i, cam)

    def handle_co_attn_query(self, block):
        cam_q_i = avg_heads(block.multihead_attn.get_attn().detach(), block.multihead_attn.get_attn_gradients().detach())
        self.R_q_i = apply_mm_attention_rules(self.R_q_q, self.R_i_i, cam_q_i,
                                              apply_normalization=self.normalize_self_attention)

    def generate_ours_abl(self, img, target_index, index=None, use_lrp=False, normalize_self_attention=False,
                          apply_self_in_rule_10=True):
        if use_lrp:
            raise NotImplementedError("use_lrp=True needs the reference's LRP layer library: out of scope")
        self.use_lrp = use_lrp
        self.normalize_self_attention = normalize_self_attention
        outputs = _logits_for_backward(self.model, img)
        _one_hot_backward(self.model, outputs, target_index, index)
        decoder_blocks = self.model.transformer.decoder.layers
        encoder_blocks = self.model.transformer.encoder.layers
        ref = encoder_blocks[0].self_attn.get_attn()
        image_bboxes = ref.shape[-1]
        queries_num = decoder_blocks[0].self_attn.get_attn().shape[-1]
        self.R_i_i = torch.eye(image_bboxes, image_bboxes, device=ref.device)
        self.R_q_q = torch.eye(queries_num, queries_num, device=ref.device)
        self.R_q_i = torch.zeros(queries_num, image_bboxes, device=ref.device)
        self.handle_self_attention_image(encoder_blocks)
        for blk in decoder_blocks:
            self.handle_co_attn_self_query(blk)
            self.handle_co_attn_query(blk)
        aggregated = self.R_q_i.unsqueeze_(0)
        return aggregated[:, target_index, :].unsqueeze_(0).detach()
