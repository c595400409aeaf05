"""Perturbation evaluator inner loop for VisualBERT --
``VisualBERT/mmf/trainers/core/evaluation_loop.py:100-166``, same scheme as ``lxmert_perturbation``:

the reference re-runs the model 9 times per sample, each time after a host ``topk`` and a physical gather of the kept
image regions / question tokens.  Here the 9 perturbed inputs are ONE batch through ``VisualBERTForClassification``:

  * image test: the visual tokens' position / type embeddings do not depend on their index
    (``embeddings.py:399-417``: all visual position ids are 0), so a region masked as an attention key is
    indistinguishable from a removed one; the step that keeps no region needs no special case here because the text
    keys stay unmasked in the same softmax;
  * text test: kept token ids are gathered and left-aligned (positions re-index exactly as after the reference's
    gather), the tail is masked; [CLS] (0), the token at ``cls_index = len - 2`` (the '?' the 'vqa' pooler reads) and
    [SEP] always stay (``evaluation_loop.py:139-143``).
"""
from __future__ import annotations

import torch

from .lxmert_perturbation import PERT_STEPS, image_keep_masks, ranking

# evaluation_loop.py:93-96: the image test removes regions on a finer scale near "all removed" than the text test
IMAGE_STEPS = (0, 0.5, 0.75, 0.95, 0.96, 0.97, 0.98, 0.99, 1)
TEXT_STEPS = PERT_STEPS


def text_keep_batch(input_ids, segment_ids, text_scores, n_text, steps=PERT_STEPS):
    """``input_ids``, ``segment_ids``: ``[1, T]`` (first ``n_text`` positions real); ``text_scores [n_text - 3]``: the
    scores of tokens ``1 .. cls_index - 1``.  Returns left-aligned ``(ids [S, T], segments [S, T], input_mask [S, T])``."""
    T = input_ids.shape[1]
    cls_index = n_text - 2
    n_inner = text_scores.shape[-1]
    order = ranking(text_scores) + 1
    S = len(steps)
    keep = torch.zeros(S, T, dtype=torch.bool, device=input_ids.device)
    keep[:, 0] = keep[:, cls_index] = keep[:, cls_index + 1] = True
    for s, step in enumerate(steps):
        keep[s, order[: int((1 - step) * n_inner)]] = True
    pos = torch.arange(T, device=input_ids.device).expand(S, T)
    perm = torch.argsort(torch.where(keep, pos, pos + T), dim=1)
    mask = torch.gather(keep, 1, perm)
    ids = torch.gather(input_ids.expand(S, T), 1, perm) * mask
    seg = torch.gather(segment_ids.expand(S, T), 1, perm) * mask
    return ids, seg, mask.long()


class VisualBertPerturbation:
    """``model``: a ``visualbert_model.VisualBERT``.  ``sample``: ``input_ids``, ``input_mask``, ``segment_ids``
    (``[1, T]``, padding allowed), ``image_feature_0`` (``[1, V, dim]``).  ``method_cam``: the generator's ``[1, N]`` row
    (``N = n_text + V``).  Both methods return the 9 steps' ``scores [S, num_labels]``.  ``steps=None`` (default): the
    reference's step lists, which differ per modality (``IMAGE_STEPS`` / ``TEXT_STEPS``, evaluation_loop.py:93-96)."""

    def __init__(self, model, steps=None):
        self.model = model
        self.image_steps = tuple(IMAGE_STEPS if steps is None else steps)
        self.text_steps = tuple(TEXT_STEPS if steps is None else steps)

    def _run(self, ids, seg, input_mask, feats, visual_mask):
        S = ids.shape[0]
        attention_mask = torch.cat((input_mask, visual_mask), dim=-1)
        return self.model.model(ids, input_mask, attention_mask, seg, feats.expand(S, -1, -1),
                                torch.zeros_like(visual_mask))["scores"]

    @torch.no_grad()
    def perturbation_image(self, sample, method_cam, is_positive_pert=False):
        n_text = int(sample["input_mask"].sum())                      # one host read per sample, as in the reference
        S = len(self.image_steps)
        keep = image_keep_masks(method_cam[0, n_text:], self.image_steps, is_positive_pert).long()    # [S, V]
        ids = sample["input_ids"][:, :n_text].expand(S, -1)
        seg = sample["segment_ids"][:, :n_text].expand(S, -1)
        return self._run(ids, seg, torch.ones_like(ids), sample["image_feature_0"], keep)

    @torch.no_grad()
    def perturbation_text(self, sample, method_cam, is_positive_pert=False):
        n_text = int(sample["input_mask"].sum())
        cam = -method_cam if is_positive_pert else method_cam
        ids, seg, mask = text_keep_batch(sample["input_ids"][:, :n_text], sample["segment_ids"][:, :n_text],
                                         cam[0, 1:n_text - 2], n_text, self.text_steps)
        feats = sample["image_feature_0"]
        visual_mask = torch.ones(len(self.text_steps), feats.shape[1], dtype=torch.long, device=feats.device)
        return self._run(ids, seg, mask, feats, visual_mask)

    @staticmethod
    def accuracy(scores, targets):
        """``targets [num_labels]``: the item's soft scores per answer (``report['targets'][0]``) -> ``[S]``."""
        return targets[scores.argmax(dim=-1)]


def evaluation_loop(generate, pert, loader, num_samples, modality="image", is_positive_pert=False, reference_exact=False):
    """``TrainerEvaluationLoopMixinPert.evaluation_loop`` (evaluation_loop.py:73-169) on the batched evaluator: for every
    item ``method_cam = generate(item)``, the 9 perturbed re-runs in ONE forward, ``step_acc[s] += targets[argmax]``.
    Returns the per-step accuracies in percent (what the reference prints) as a ``[9]`` fp64 tensor.

    ``reference_exact=True`` reproduces the reference's sample accounting: its ``i > num_samples`` test stops only AFTER
    item ``num_samples + 1`` and it still divides by ``num_samples``.  The default evaluates exactly ``num_samples`` items.
    ``loader`` yields ``sample_list`` dicts carrying ``targets [1, num_labels]``."""
    steps = pert.image_steps if modality == "image" else pert.text_steps
    step_acc = torch.zeros(len(steps), dtype=torch.float64)
    limit = num_samples + 1 if reference_exact else num_samples
    for i, item in enumerate(loader):
        if i >= limit:
            break
        cam = generate(item).detach()
        run = pert.perturbation_image if modality == "image" else pert.perturbation_text
        scores = run(item, cam, is_positive_pert)
        step_acc += pert.accuracy(scores, item["targets"][0]).double().cpu()
    return step_acc / num_samples * 100
