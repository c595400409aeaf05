"""pytest config: registers the ``gpu`` marker and makes the repo root importable.

``-m "not gpu"`` runs here (no GPU): oracle vs golden vectors, host logic, C-ABI symbol checks,
gloo world_size-2 sharding tests.  ``-m gpu`` runs on an MI355X and calls the HIP path through the C-ABI.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (HIP kernels are executed)")
    # debugging aid: MMX_TEST_TOGGLES="norules,novalue,norecord,nosplit,nolxmert" switches schedule features off for a whole run
    toggles = [t for t in os.environ.get("MMX_TEST_TOGGLES", "").split(",") if t]
    if toggles:
        import torch
        from transformer_mm_explainability_amd import attention_modules, detr_explainability, lxmert_model, ops
        if "norules" in toggles:
            detr_explainability.Generator.overlap_rules = False
        if "novalue" in toggles:
            attention_modules.MultiheadAttention.overlap_value_proj = False
        if "nolxmert" in toggles:
            lxmert_model.LxmertEncoder.overlap_modalities = False
        if "norecord" in toggles:
            torch.Tensor.record_stream = lambda self, stream: None
        if "nosplit" in toggles:
            ops.set_option("attn_fwd_split", 0)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    return load


def pytest_sessionfinish(session, exitstatus):
    """Largest absolute error every ``tests/parity.close`` comparison saw (GPU suites) -> gpurun_out/parity_errors.json."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        import parity
        parity.dump(os.path.join(ROOT, "gpurun_out", "parity_errors.json"))
    except Exception:           # a read-only tree must not fail the session
        pass
