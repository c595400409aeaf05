"""-m gpu: on-device heat-map and Otsu-mask kernels against the oracle (torch CPU interpolate; OpenCV Otsu restated)."""
import numpy as np
import pytest
import torch

from oracle import relevancy_np as onp

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("g,size", [(7, 224), (14, 224), (24, 336), (2, 9)])
def test_heatmaps(g, size):
    from transformer_mm_explainability_amd import postprocess
    gen = torch.Generator().manual_seed(g)
    rel = torch.rand(5, g * g, generator=gen) + torch.arange(5)[:, None]
    got = postprocess.image_heatmaps(rel.cuda(), size).cpu().numpy()
    for b in range(5):
        np.testing.assert_allclose(got[b], onp.heatmap_bilinear_minmax(rel[b].numpy(), size), rtol=1e-5, atol=2e-6)
    one = postprocess.image_heatmaps(rel[0].cuda(), size)
    assert one.shape == (size, size)


def test_otsu_masks():
    from transformer_mm_explainability_amd import postprocess
    gen = torch.Generator().manual_seed(3)
    h, w = 25, 38
    cams = torch.rand(6, 1, h * w, generator=gen) ** 3            # skewed: bimodal-ish histogram
    cams[2] = torch.cat([torch.rand(1, 400, generator=gen) * 0.1, torch.rand(1, h * w - 400, generator=gen) * 0.1 + 0.8], 1)
    masks, thr = postprocess.otsu_masks(cams.cuda(), return_thresholds=True)
    assert masks.shape == cams.shape
    for k in range(6):
        want, t = onp.otsu_mask(cams[k].numpy())
        assert int(thr[k]) == t
        np.testing.assert_array_equal(masks[k].cpu().numpy(), want)
