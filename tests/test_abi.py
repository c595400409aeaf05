"""CPU: the C-ABI library builds, loads, and exports every symbol include/mmx_relevancy.h declares (no compute calls);
argument validation that needs no GPU returns the documented error codes."""
import ctypes as C
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from transformer_mm_explainability_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "transformer-mm-explainability_amd", "csrc"), "-j4"], check=True,
                       capture_output=True)
    return _lib


def test_header_symbols_are_exported_and_bound(lib):
    handle = lib.lib()
    declared = lib.header_symbols()
    assert len(declared) >= 20
    missing = [s for s in declared if not hasattr(handle, s)]
    assert not missing, missing
    assert set(lib._PROTOTYPES) == set(declared), set(lib._PROTOTYPES) ^ set(declared)
    assert handle.mmx_abi_version() == 1


def test_argument_validation_without_gpu(lib):
    handle = lib.lib()
    null = C.c_void_p(0)
    assert handle.mmx_avg_heads(null, null, null, 1, 1, 1, 1, 0, null) == -22           # MMX_EINVAL: null pointers
    assert b"null" in handle.mmx_last_error()
    assert handle.mmx_set_option(b"no_such_option", 1) == -22
    assert handle.mmx_set_option(b"self_chain_groups", 0) == 0
    assert handle.mmx_self_chain_workspace_bytes(12, 64, 8, 300, 0, 0) > 0              # split path needs scratch
    assert handle.mmx_mm_rules_workspace_bytes(20, 36) > 0


def test_no_cpu_fallback():
    """CPU tensors are refused loudly; the package never imports the oracle."""
    import torch
    from transformer_mm_explainability_amd import ops
    from transformer_mm_explainability_amd._lib import MMXError
    with pytest.raises(MMXError):
        ops.avg_heads(torch.rand(2, 4, 4), torch.rand(2, 4, 4))
    pkg = os.path.join(ROOT, "transformer-mm-explainability_amd")
    for name in os.listdir(pkg):
        if name.endswith(".py"):
            src = open(os.path.join(pkg, name)).read()
            assert "import oracle" not in src and "from oracle" not in src, name


def test_tuned_gemm_files_are_tunableop_selections():
    """``tuned_gemms.WORKLOADS`` files: present, TunableOp CSVs for gfx950 (validator rows first, then op,shape,solution,time)."""
    import importlib
    tg = importlib.import_module("transformer_mm_explainability_amd.tuned_gemms")
    for workload, name in tg.WORKLOADS.items():
        assert tg.available(workload), workload
        rows = [l.strip().split(",") for l in open(os.path.join(os.path.dirname(tg.__file__), "tuning", name)) if l.strip()]
        validators = [r for r in rows if r[0] == "Validator"]
        assert any(r[1] == "GCN_ARCH_NAME" and r[2].startswith("gfx950") for r in validators), name
        entries = [r for r in rows if r[0] != "Validator"]
        assert entries and all(len(r) >= 4 and "TunableOp" in r[0] and float(r[-1]) > 0 for r in entries), name
