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
    assert handle.mmx_abi_version() == 2 == lib.ABI_VERSION


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


_ASAN_SWEEP = r"""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
from transformer_mm_explainability_amd import _lib
h = _lib.lib()
assert _lib.LIB_PATH.endswith("libmmx_hip_asan.so") and hasattr(C.CDLL(None), "__asan_init"), "sanitizer runtime not loaded"
swept = 0
for name, (res, args) in sorted(_lib._PROTOTYPES.items()):                 # 1. every entry point with all-null / all-zero arguments
    if name in ("mmx_last_error", "mmx_abi_version"):
        continue
    zeros = [None if a in (C.c_void_p, C.POINTER(C.c_void_p), C.c_char_p, C.POINTER(C.c_float)) else (0.0 if a is C.c_float else 0)
             for a in args]
    rc = getattr(h, name)(*zeros)
    swept += 1
    if res is C.c_int and args and name != "mmx_set_option":
        assert rc < 0, (name, rc)                                          # refused with a status, never a crash
        assert h.mmx_last_error()
for key in (b"self_chain_algo", b"self_chain_groups", b"self_chain_pipe", b"self_chain_nt", b"bmm_tiles", b"attn_head", b"debug_flags",
            b"", b"x" * 4096):                                              # 2. option parsing incl. out-of-range values / odd keys
    for value in (-1, 0, 1, 5, 99, 2 ** 31 - 1):
        h.mmx_set_option(key, value)
for key, value in ((b"self_chain_algo", 0), (b"self_chain_groups", 0), (b"self_chain_pipe", 4), (b"self_chain_nt", 1), (b"bmm_tiles", 1),
                   (b"attn_head", 1), (b"debug_flags", 0)):
    assert h.mmx_set_option(key, value) == 0
for L in (1, 12, 48):                                                       # 3. workspace queries over the shape space
    for B in (1, 64, 1024):
        for N in (1, 50, 77, 128, 129, 577, 1050):
            for M in (0, 100):
                h.mmx_self_chain_workspace_bytes(L, B, 8, N, M, 0)
    h.mmx_rollout_workspace_bytes(B, 577)
h.mmx_lxmert_schedule_workspace_bytes(32, 12, 14, 36, 9, 5)
h.mmx_detr_decoder_rows_workspace_bytes(10, 8, 100, 950)
if sys.argv[1] == "nogpu":
    # 4. no device in this process: launches fail inside the HIP runtime, so the host side of a full call can run on made-up device
    #    addresses -- pointer tables of the maximum length, ChainArgs set-up, the dispatcher -- and must come back with a status
    fake = (C.c_void_p * 48)(*[0x7f0000000000 + 4096 * i for i in range(48)])
    tbl = C.cast(fake, C.POINTER(C.c_void_p))
    ws = C.c_void_p(0x7e0000000000)
    for (L, B, H, N) in ((48, 64, 8, 77), (12, 64, 12, 50), (48, 2, 16, 577), (49, 1, 1, 8), (3, 1, 1, 7)):
        need = h.mmx_self_chain_workspace_bytes(min(L, 48), B, H, N, 0, 0)
        rc = h.mmx_relevancy_self_chain(tbl, tbl, L, B, H, N, 0, None, ws, None, None, 0, ws, need, None)
        assert rc < 0, (L, B, H, N, rc)
    rc = h.mmx_avg_heads(ws, ws, ws, 4, 8, 77, 77, 0, None)
    assert rc < 0
print("ASAN_SWEEP_OK", swept)
"""


def test_argument_validation_under_host_asan():
    """The host side of the C-ABI under AddressSanitizer (``make -C csrc -f Makefile.asan asan-host``: host code instrumented, gfx950 code objects as
    in the product build): every entry point with null / zero arguments, option parsing, the workspace queries over the shape
    space, and -- where no GPU is visible -- full calls on made-up device addresses, which exercise the pointer tables, ``ChainArgs``
    and the dispatcher up to the (failing) launch.  Any heap / stack / global overflow or use-after-free in the wrappers aborts the
    child with a report.  GPU-side ASAN (XNACK-on code objects and runtime mode) is refused by the GPU pool: tests/test_gpu_stress.py
    covers the device side from the outside."""
    import subprocess
    import sys
    csrc = os.path.join(ROOT, "transformer-mm-explainability_amd", "csrc")
    so, rt = os.path.join(csrc, "asan", "libmmx_hip_asan.so"), os.path.join(csrc, "asan", "runtime_path.txt")
    # (make is incremental: a no-op when the sanitizer build is current, a rebuild of what changed otherwise)
    build = subprocess.run(["make", "-C", csrc, "-f", "Makefile.asan", "-j4", "asan-host"], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-2000:]
    runtime = open(rt).read().strip()
    assert os.path.exists(runtime), runtime
    try:
        import torch
        nogpu = not torch.cuda.is_available()
    except Exception:
        nogpu = False
    env = dict(os.environ, LD_PRELOAD=runtime, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0", MMX_LIB_PATH=so)
    out = subprocess.run([sys.executable, "-c", _ASAN_SWEEP, "nogpu" if nogpu else "gpu"], cwd=ROOT, capture_output=True, text=True,
                         timeout=600, env=env)
    assert out.returncode == 0 and "ASAN_SWEEP_OK" in out.stdout, (out.stdout[-1500:], out.stderr[-4000:])
    assert "AddressSanitizer" not in out.stderr, out.stderr[-4000:]
