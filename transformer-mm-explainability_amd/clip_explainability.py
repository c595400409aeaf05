"""Drop-in ``interpret`` for CLIP -- same signatures as the reference, relevancy chain on the fused HIP kernel.

Reference entry points mirrored here:
  * ``interpret(image, texts, model, device, start_layer=-1, start_layer_text=-1)``
    -- CLIP_explainability.ipynb cell 6 (batched: one image, B texts) -> ``(text_relevance [B,Nt,Nt], image_relevance [B,Ni-1])``
  * ``interpret_single(image, text, model, device, index=None)``
    -- CLIP/example.py:8-32 (one image, K texts, explains ``logits_per_image[0, index]``) -> ``image_relevance [Ni-1]``
  * ``text_scores(text_encoding, R_text)`` -- notebook cell 8:5-7 post-processing (on device)

What is different under the hood (results agree to fp32 rounding, see tests/test_gpu_clip.py):
  * the reference issues one ``torch.autograd.grad`` per layer (24 partial backward passes); here ONE backward
    fills every layer's gradient slab (``capture_only`` also drops weight/input gradients);
  * the per-layer reshape/mul/clamp/mean/bmm/add launches are one ``relevancy_self_chain`` launch per tower.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops

class _Frozen:
    """Temporarily mark parameters as not requiring grad: the explainability pass needs d(logit)/d(probs) only."""

    def __init__(self, model):
        self.params = [p for p in model.parameters() if p.requires_grad]

    def __enter__(self):
        for p in self.params:
            p.requires_grad_(False)

    def __exit__(self, *exc):
        for p in self.params:
            p.requires_grad_(True)


def _chains(model, batch_size, start_layer, start_layer_text):
    vis, txt = model.visual.transformer, model.transformer
    if start_layer == -1:
        start_layer = vis.layers - 1
    if start_layer_text == -1:
        start_layer_text = txt.layers - 1
    vb, tb = vis.buffers, txt.buffers
    # Same stream, back to back: each launch already fills the chip (layer-group split), and measured on MI355X two
    # concurrent chain kernels on two streams are slower than the two in sequence (profiles/r01_chain_probe.txt).
    R_text = _plan(tb, start_layer_text, txt.layers, batch_size, False).launch()
    R = _plan(vb, start_layer, vis.layers, batch_size, vb.shared_probs and batch_size > 1).launch()
    return R_text, R


def _plan(buffers, first, last, batch_size, shared):
    """Prepared chain launch, cached on the slab object (the slabs keep their addresses from call to call)."""
    cache = buffers.__dict__.setdefault("_chain_plans", {})
    key = (first, last, batch_size, shared)
    if key not in cache:
        cache[key] = ops.ChainPlan([buffers.probs[l] for l in range(first, last)],
                                   [buffers.grads[l] for l in range(first, last)], batch_size, shared_attn=shared)
    return cache[key]


def interpret(image, texts, model, device, start_layer=-1, start_layer_text=-1, share_image_forward=True):
    """CLIP_explainability.ipynb cell 6.  ``image``: ``[1,3,R,R]``, ``texts``: ``[B, context]`` token ids.

    ``share_image_forward`` (extra keyword, default on): the reference repeats the ONE image B times (cell 6:3) and
    runs the image tower on B identical copies.  Here its forward runs once and only the backward -- whose upstream
    gradients do differ per text -- runs at batch B (``clip_model.Transformer.forward_shared``); results are the
    same to fp32 rounding (``tests/test_gpu_clip.py``).  ``False`` runs the B copies like the reference.
    """
    batch_size = texts.shape[0]
    sl = model.visual.transformer.layers - 1 if start_layer == -1 else start_layer
    slt = model.transformer.layers - 1 if start_layer_text == -1 else start_layer_text
    prev = (model.capture_only, model.first_grad_layers)
    model.capture_only, model.first_grad_layers = True, (sl, slt)    # no gradient work below the start layers
    try:
        with _Frozen(model), torch.enable_grad():
            eye = torch.eye(batch_size, dtype=torch.float32, device=texts.device)
            if share_image_forward and image.shape[0] == 1 and batch_size > 1:
                feat1, state = model.visual.forward_shared(image.type(model.dtype), batch_size)
                image_features = feat1.expand(batch_size, -1).contiguous().requires_grad_(True)   # per-sample leaf
                logits_per_image, _ = model.logits(image_features, model.encode_text(texts))
                torch.autograd.backward(logits_per_image, grad_tensors=eye)
                model.visual.backward_shared(state, image_features.grad, sl)
            else:
                images = image.repeat(batch_size, 1, 1, 1)
                logits_per_image, _ = model(images, texts)
                # one_hot = sum_i logits_per_image[i, i]  (cell 6:6-10)  ->  d one_hot / d logits = I
                torch.autograd.backward(logits_per_image, grad_tensors=eye)
    finally:
        model.capture_only, model.first_grad_layers = prev
    R_text, R = _chains(model, batch_size, start_layer, start_layer_text)
    image_relevance = R[:, 0, 1:]
    return R_text, image_relevance


def interpret_single(image, text, model, device, index=None):
    """CLIP/example.py:8-32 without the plotting: returns ``image_relevance [Ni-1]`` (``R[0,0]`` zeroed first)."""
    prev = (model.capture_only, model.first_grad_layers)
    model.capture_only, model.first_grad_layers = True, (0, model.transformer.layers)   # text tower: no gradient work
    try:
        with _Frozen(model), torch.enable_grad():
            logits_per_image, _ = model(image, text)
            if index is None:
                index = int(np.argmax(logits_per_image.detach().cpu().numpy(), axis=-1).reshape(-1)[0])
            one_hot = torch.zeros_like(logits_per_image)
            one_hot[0, index] = 1
            torch.autograd.backward(logits_per_image, grad_tensors=one_hot)
    finally:
        model.capture_only, model.first_grad_layers = prev
    vis = model.visual.transformer
    vb = vis.buffers
    # example.py flattens batch*heads into one head axis (cam.reshape(-1, N, N).mean(0)): batch_size = 1 here
    R = ops.relevancy_self_chain([vb.probs[l] for l in range(vis.layers)],
                                 [vb.grads[l] for l in range(vis.layers)], 1)[0]
    R[0, 0] = 0
    return R[0, 1:]


def text_scores(text_encoding, R_text):
    """Notebook cell 8:5-7 for one sample: row of the EOT token, columns ``1:EOT``, normalised by their sum."""
    cls_idx = int(text_encoding.argmax(dim=-1))
    row = R_text[cls_idx, 1:cls_idx]
    return (row / row.sum()).flatten()


def image_heatmap(image_relevance, size=224):
    """Notebook cell 7:14-18: reshape to the patch grid, bilinear upsample, min-max normalise -> ``[size, size]``
    (``postprocess.image_heatmaps`` does a whole batch in one launch)."""
    from .postprocess import image_heatmaps
    return image_heatmaps(image_relevance.reshape(-1), size)
