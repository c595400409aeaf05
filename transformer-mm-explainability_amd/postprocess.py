"""On-device post-processing of relevancy maps: the reference moves every map to the host
for these steps (numpy min-max, ``cv2.threshold``), i.e. one synchronisation per map inside the evaluator loops."""
from __future__ import annotations

import torch

from . import ops
from ._lib import MMXError, check, lib


def image_heatmaps(image_relevance, size=224):
    """``[B, P]`` (or ``[P]``) patch relevancies (``P = g*g``) -> ``[B, size, size]`` heat maps: bilinear upsample
    (``interpolate(..., mode='bilinear')``) + min-max normalisation, one kernel launch for the whole batch.
    Reference: CLIP_explainability.ipynb cell 7:14-18 (``size=224``), ViT notebook cell 8:25-28 (``scale_factor=16``)."""
    squeeze = image_relevance.dim() == 1
    rel = image_relevance.reshape(1, -1) if squeeze else image_relevance
    ops._dev(rel)
    rel = ops._f32c(rel)
    B, P = rel.shape
    g = int(round(P ** 0.5))
    if g * g != P:
        raise MMXError("image_heatmaps: %d patches are not a square grid" % P)
    out = torch.empty(B, size, size, dtype=torch.float32, device=rel.device)
    check(lib().mmx_heatmap_bilinear_minmax(ops._p(rel), ops._p(out), B, g, size, ops._stream()),
          "mmx_heatmap_bilinear_minmax")
    return out[0] if squeeze else out


def otsu_masks(cams, return_thresholds=False):
    """``[K, ...]`` relevancy maps (one per kept query) -> ``[K, ...]`` fp32 masks in {0, 255}: min-max to [0, 255],
    8-bit truncation, Otsu threshold, ``THRESH_BINARY`` -- DETR/mask_generator.py:116-121 for all K queries in one launch."""
    ops._dev(cams)
    c = ops._f32c(cams).reshape(cams.shape[0], -1)
    K, n = c.shape
    masks = torch.empty(K, n, dtype=torch.float32, device=c.device)
    thr = torch.empty(K, dtype=torch.int32, device=c.device)
    check(lib().mmx_otsu_masks(ops._p(c), ops._p(masks), ops._p(thr), K, n, ops._stream()), "mmx_otsu_masks")
    masks = masks.reshape(cams.shape)
    return (masks, thr) if return_thresholds else masks
