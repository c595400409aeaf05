"""CPU suite: what "pinned" means for the Otsu step when OpenCV cannot be imported (cv2 and skimage are absent from this image;
``DETR/mask_generator.py:114-120`` calls ``cv2.threshold(..., THRESH_BINARY + THRESH_OTSU)``).

Otsu's threshold is DEFINED as the level that maximises the between-class variance of the 8-bit histogram, so the restatement
(``oracle/relevancy_np.otsu_mask``, the referee of the HIP ``otsu_kernel``) is checked against that definition by EXHAUSTIVE
search in float64 over all 256 candidate levels, plus OpenCV's documented tie rule (``getThreshVal_Otsu_8u``: levels are
scanned upwards and a later level replaces the current one only if its variance is STRICTLY larger -> the lowest maximiser wins)
and its ``THRESH_BINARY`` convention (``pixel > threshold -> 255``).  This is a property pin, not a golden-vector pin."""
import numpy as np
import pytest

from oracle import relevancy_np as onp


def _levels(cam):
    cam = np.asarray(cam, dtype=np.float32)
    cam = (cam - cam.min()) / (cam.max() - cam.min()) * np.float32(255)      # mask_generator.py:116-117
    return cam.astype(np.uint8)


def _between_class_variance(img):
    """sigma_b^2(t) for t = 0..255, classes {<= t} / {> t}, float64; -inf where a class is empty."""
    hist = np.bincount(img.reshape(-1), minlength=256).astype(np.float64)
    p = hist / hist.sum()
    lv = np.arange(256, dtype=np.float64)
    out = np.full(256, -np.inf)
    for t in range(256):
        q1, q2 = p[:t + 1].sum(), p[t + 1:].sum()
        if q1 <= 0 or q2 <= 0:
            continue
        mu1, mu2 = (lv[:t + 1] * p[:t + 1]).sum() / q1, (lv[t + 1:] * p[t + 1:]).sum() / q2
        out[t] = q1 * q2 * (mu1 - mu2) ** 2
    return out


def _cases():
    g = np.random.default_rng(7)
    yield "bimodal", np.concatenate([g.normal(0.2, 0.05, 600), g.normal(0.8, 0.07, 350)]).reshape(25, 38)
    yield "skewed", g.gamma(2.0, 1.0, (25, 38))
    yield "uniform", g.random((25, 42))
    yield "relevancy-like", np.abs(g.standard_cauchy((25, 34))) * 1e-4
    yield "three levels", g.choice([0.0, 0.4, 1.0], size=(20, 30), p=[0.5, 0.3, 0.2])
    yield "two levels", g.choice([0.0, 1.0], size=(10, 10), p=[0.7, 0.3])
    yield "symmetric tie", np.array([[0.0, 0.0, 1.0, 1.0, 0.5, 0.5, 0.25, 0.75]])
    for k in range(20):
        yield "random %d" % k, g.random((25, 38)) ** (1 + k % 4)


@pytest.mark.parametrize("name,cam", list(_cases()), ids=[n for n, _ in _cases()])
def test_oracle_threshold_maximises_between_class_variance(name, cam):
    mask, thr = onp.otsu_mask(cam)
    img = _levels(cam)
    sigma = _between_class_variance(img)
    best = sigma.max()
    assert np.isfinite(best)
    # the chosen level is a maximiser (fp64; OpenCV's running-mean recurrence differs from the direct sums by rounding only)
    assert sigma[thr] >= best * (1 - 1e-12), (thr, int(sigma.argmax()), sigma[thr], best)
    # tie rule: the LOWEST level whose variance equals the maximum (strict '>' in the upward scan)
    ties = np.nonzero(sigma >= best * (1 - 1e-12))[0]
    assert thr == ties[0], (thr, ties)
    # THRESH_BINARY: strictly above the threshold -> 255
    np.testing.assert_array_equal(mask, np.where(img > thr, 255, 0).astype(np.float32))
