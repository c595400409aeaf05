"""Pre-tuned library-GEMM selections (PyTorch-ROCm TunableOp) for the model bodies of this package on gfx950.

The bodies' GEMMs are plain library GEMMs (hipBLASLt / rocBLAS); for the small and skinny shapes of the explainability
passes (DETR decoder at 100 queries, LXMERT at 20 + 36 tokens, batch-1 shared forwards) the default heuristic often picks
a solution that runs on ONE workgroup (e.g. ``MT256x112x32`` for a ``[100, 256] x [256, 256]`` product: 31 us on 1 of 256
CUs).  ``tuning/*.csv`` hold the solutions TunableOp selected on an MI355X for each workload (made by
``tools/tune_gemms.py``); ``enable(workload)`` turns TunableOp on with tuning OFF and loads that file, so shapes that are
in it use the tuned solution and everything else the default.  PyTorch ignores a file whose validators (ROCm / hipBLASLt
version, architecture) do not match the box.  Process-wide (it changes every GEMM of the process), hence opt-in.
"""
import contextlib
import os
import threading

import torch

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning")
# (LXMERT: a selection was made too, but with it the capture of GraphedGenerateOursBatch died with
#  hipErrorStreamCaptureUnsupported inside a library call -- not shipped; `tools/tune_gemms.py lxmert` reproduces it.)
#  Rounds 3-5 also shipped a selection for the perturbation re-runs ("lxmert_pert", used through ``scope``).  Round 6 REMOVED it:
#  at the evaluator's text-test shapes (5760 / 10368 rows) one of its solutions never finishes on the GPU -- a hang, found when the
#  evaluator's ``--text`` run was exercised at full size -- and since the re-runs are replayed from a hipGraph (default selection) or
#  are host-bound when eager, it bought nothing any more.  ``scope("lxmert_pert")`` is now a no-op unless a file is regenerated.)
WORKLOADS = {"clip_vitb32_b64": "tunableop_gfx950_clip_vitb32_b64.csv", "detr": "tunableop_gfx950_detr_r50.csv",
             "clip_vitl14_336_bf16": "tunableop_gfx950_clip_vitl14_336_bf16.csv"}      # ("lxmert_pert": not shipped any more, see above)
_LOADED = {}


def available(workload):
    return workload in WORKLOADS and os.path.exists(os.path.join(_DIR, WORKLOADS[workload]))


def enable(workload):
    """Returns True if the tuned selection for ``workload`` was loaded (False: no file for it, nothing changed)."""
    if not available(workload):
        return False
    tun = torch.cuda.tunable
    tun.enable(True)
    tun.tuning_enable(False)
    tun.record_untuned_enable(False) if hasattr(tun, "record_untuned_enable") else None
    return bool(tun.read_file(os.path.join(_DIR, WORKLOADS[workload])))


_LOCK = threading.RLock()      # re-entrant: a helper may open its own scope inside a caller's (same thread)
_DEPTH = [0]                   # nesting depth under _LOCK: only the OUTERMOST scope saves / restores the TunableOp state


def _flag(tun, getter):
    fn = getattr(tun, getter, None)
    return fn() if fn is not None else None


@contextlib.contextmanager
def scope(workload):
    """TunableOp on, with ``workload``'s selection, for the duration of the block only; the previous state -- on / off AND the
    caller's ``tuning_enable`` / ``record_untuned_enable`` flags (a process started with ``PYTORCH_TUNABLEOP_TUNING=1`` keeps
    tuning afterwards) -- comes back on exit (yields whether the selection is in effect).  For eager passes that sit next to
    captured ones in one process: a hipGraph capture must not run into a tuned solution (some allocate inside the library call),
    so nothing is switched while the current stream is capturing.  The TunableOp state is PROCESS-global: scopes are serialised
    by a lock, and a scope must not overlap a hipGraph capture running on ANOTHER thread (this module cannot see that)."""
    if not torch.cuda.is_available() or not available(workload) or torch.cuda.is_current_stream_capturing():
        yield False
        return
    tun = torch.cuda.tunable
    with _LOCK:
        if _DEPTH[0] > 0:          # nested on this thread: the outer scope owns the saved state; just make sure the selection is on
            if workload not in _LOADED:
                _LOADED[workload] = enable(workload)
            _DEPTH[0] += 1
            try:
                yield _LOADED[workload]
            finally:
                _DEPTH[0] -= 1
            return
        # the state capture and `enable` (a file read) run INSIDE the try whose finally resets the depth: an exception in the
        # prologue must not leave _DEPTH at 1 (every later scope would take the nested branch and never restore anything)
        _DEPTH[0] = 1
        saved = None
        try:
            saved = (tun.is_enabled(), _flag(tun, "tuning_is_enabled"), _flag(tun, "record_untuned_is_enabled"))
            if workload not in _LOADED:
                _LOADED[workload] = enable(workload)
            else:
                tun.enable(True)
                tun.tuning_enable(False)
            yield _LOADED[workload]
        finally:
            _DEPTH[0] = 0
            if saved is not None:
                was, was_tuning, was_record = saved
                if was_tuning is not None:
                    tun.tuning_enable(was_tuning)
                if was_record is not None and hasattr(tun, "record_untuned_enable"):
                    tun.record_untuned_enable(was_record)
                tun.enable(was)
