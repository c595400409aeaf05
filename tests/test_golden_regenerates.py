"""CPU suite, build container only: every committed fixture under ``tests/golden/`` is REGENERABLE -- ``make_golden.py`` run
against ``/root/reference`` into a scratch directory reproduces each ``.npz`` array bit for bit (``lrp_layers.npz``
was built from the unseeded global generator and could not be regenerated).  Skipped where the reference checkout does
not exist (the GPU box)."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("MMX_REFERENCE", "/root/reference")


def _same(a, b):
    if a.dtype.kind in "fc":
        return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b, equal_nan=True)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "DETR")), reason="needs the reference checkout")
def test_every_golden_fixture_regenerates_bit_for_bit(tmp_path):
    env = dict(os.environ, MMX_GOLDEN_OUT=str(tmp_path))
    subprocess.run([sys.executable, os.path.join(HERE, "golden", "make_golden.py")], check=True, env=env,
                   stdout=subprocess.DEVNULL, timeout=600)
    committed = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))
    assert committed
    made = {os.path.basename(p) for p in glob.glob(str(tmp_path / "*.npz"))}
    assert made == {os.path.basename(p) for p in committed}, "fixture set and generator disagree"
    differ = []
    for path in committed:
        a, b = np.load(path), np.load(str(tmp_path / os.path.basename(path)))
        if set(a.files) != set(b.files) or not all(_same(a[k], b[k]) for k in a.files):
            differ.append(os.path.basename(path))
    assert not differ, "not regenerable bit for bit: %s" % differ
