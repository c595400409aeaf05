"""Module-level rule functions with the reference's exact names and signatures, running on the HIP kernels.

One module serves the three (copy-pasted) reference files; where their signatures differ the DETR / LXMERT / ViT /
VisualBERT flavours are separate names and the per-family modules re-export the right one:

  reference                                                             here
  -------------------------------------------------------------------   ---------------------------------
  avg_heads(cam, grad)                       (all three files)          avg_heads
  apply_self_attention_rules(R_ss, R_sq, cam_ss)   DETR / LXMERT        apply_self_attention_rules
  apply_self_attention_rules(R_ss, cam_ss)         ViT notebook         apply_self_attention_rules_vit
  apply_mm_attention_rules(R_ss, R_qq, cam_sq, ...)       DETR 5-arg    apply_mm_attention_rules_detr
  apply_mm_attention_rules(R_ss, R_qq, R_qs, cam_sq, ...) LXMERT 6-arg  apply_mm_attention_rules_lxmert
  handle_residual(orig_self_attention)                                  handle_residual
  compute_rollout_attention(mats, start_layer=0)   DETR / LXMERT        compute_rollout_attention
  compute_rollout_attention(mats, start_layer=0)   VisualBERT           compute_rollout_attention_batched

Inputs must live on the MI355X; there is no CPU path (``ops`` raises).
"""
from __future__ import annotations

import torch

from . import ops


class frozen_parameters:
    """``with frozen_parameters(model):`` -- parameters do not require grad inside the block.

    The explainability backward needs d(logit)/d(attention probabilities) only; the reference lets autograd also
    compute every weight gradient and never reads them.  Freezing the parameters while the forward graph is built
    removes those GEMMs (measured on the DETR-R50 head in DESIGN.md section 5).
    """

    def __init__(self, model):
        self.params = [p for p in model.parameters() if p.requires_grad]

    def __enter__(self):
        for p in self.params:
            p.requires_grad_(False)
        return self

    def __exit__(self, *exc):
        for p in self.params:
            p.requires_grad_(True)


def forward_for_backward(model, run):
    """``run()`` -> the score tensor the one-hot backward starts from, computed under frozen parameters if possible.

    Bodies built on the capture op stay differentiable with every parameter frozen (an attention block whose inputs
    are graph constants ties itself to the graph, ``capture.attention_capture``), so the backward pass is activation
    gradients only.  Any other body (the reference's hooked modules, a test double) may come out detached; then the
    forward is simply repeated the reference's way, parameters requiring grad.
    """
    if isinstance(model, torch.nn.Module):
        with frozen_parameters(model):
            out = run()
        if out.requires_grad:
            return out
    return run()


def avg_heads(cam, grad):
    """Rule 5 (DETR/modules/ExplanationGenerator.py:19-24): ``(grad*cam).clamp(min=0).mean(dim=0)`` over all leading dims."""
    return ops.avg_heads(cam, grad, batch_size=1)[0]


def apply_self_attention_rules(R_ss, R_sq, cam_ss):
    """Rules 6+7 (DETR/...:27-30, lxmert/...:26-29) -> ``(R_ss_addition, R_sq_addition)``."""
    return ops.matmul(cam_ss, R_ss), ops.matmul(cam_ss, R_sq)


def apply_self_attention_rules_vit(R_ss, cam_ss):
    """Rule 6, 2-argument ViT form (ViT notebook cell 7:10-12)."""
    return ops.matmul(cam_ss, R_ss)


def handle_residual(orig_self_attention):
    """Eq. 8-9 (DETR/...:46-53); raises AssertionError when ``diag(R - I).min() < 0`` like the reference."""
    return ops.handle_residual(orig_self_attention, check_diag=True)


def apply_mm_attention_rules_detr(R_ss, R_qq, cam_sq, apply_normalization=True, apply_self_in_rule_10=True):
    """Rule 10, DETR form (DETR/...:33-43): NaNs of the addition are zeroed."""
    return ops.mm_attention_rules(R_ss, R_qq, cam_sq, None, apply_normalization, apply_self_in_rule_10,
                                  nan_to_zero=True)


def apply_mm_attention_rules_lxmert(R_ss, R_qq, R_qs, cam_sq, apply_normalization=True, apply_self_in_rule_10=True):
    """Rules 10+11, LXMERT form (lxmert/...:32-42) -> ``(R_sq_addition, R_ss_addition)``; NaNs propagate."""
    return ops.mm_attention_rules(R_ss, R_qq, cam_sq, R_qs, apply_normalization, apply_self_in_rule_10,
                                  nan_to_zero=False)


def compute_rollout_attention(all_layer_matrices, start_layer=0):
    """DETR/...:5-16 == lxmert/...:5-15: add I, row-normalise, left-multiply from ``start_layer``.
    Matrices ``[1, N, N]`` (or ``[N, N]``); returns the shape of one input matrix."""
    mats = list(all_layer_matrices)[start_layer:]
    out = ops.rollout_chain(mats, normalize=True)
    return out.reshape(mats[0].shape)


def compute_rollout_attention_batched(all_layer_matrices, start_layer=0):
    """VisualBERT/.../ExplanationGenerator.py:5-17: batched ``[B, N, N]``, add I, NO normalisation, ``bmm`` chain."""
    mats = list(all_layer_matrices)[start_layer:]
    return ops.rollout_chain(mats, normalize=False)


def gradcam(cam, grad):
    """``Generator.gradcam`` (DETR/...:275-280, lxmert/...:542-547): ``(cam * grad.mean([1,2])).mean(0).clamp(min=0)``.
    Tiny reduction epilogue of a baseline method; kept on stock device ops."""
    cam = cam.reshape(-1, cam.shape[-2], cam.shape[-1])
    grad = grad.reshape(-1, grad.shape[-2], grad.shape[-1])
    grad = grad.mean(dim=[1, 2], keepdim=True)
    return (cam * grad).mean(0).clamp(min=0)
