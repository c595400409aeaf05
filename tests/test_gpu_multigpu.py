"""-m gpu, needs >= 2 visible GPUs (skipped on the 1-GPU boxes the suite normally runs on): the N > 1 paths over RCCL --
``bench.py --gpus 2`` launched the way the driver launches it, and the sharded LXMERT perturbation evaluator with two ranks
vs one rank (same sample list, rank-strided shards, one all-gather: identical step accuracies).  The same code paths run
on CPU with gloo in tests/test_sharding_gloo.py / test_evaluators_gloo.py."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _need_two_gpus():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL refuses two ranks on one device)")


def _launch(script_args, nproc, port, **env):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(port)] + script_args
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]            # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_bench_two_ranks_over_rccl():
    _need_two_gpus()
    line = _launch(["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--headline-only"], 2,
                   29500 + os.getpid() % 400)
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["global_batch"] == 128
    assert abs(line["value"] - 128 / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-3


def test_bench_two_ranks_on_one_gpu_over_gloo():
    """The N > 1 code path of ``bench.py`` on a 1-GPU box: two ranks share ``cuda:0`` (``MMX_BENCH_SHARE_DEVICE``) and the
    exchange step runs over gloo (``MMX_BENCH_BACKEND``; RCCL refuses two ranks per device) -- barriers, the max over ranks, the
    gather of every rank's maps, ONE JSON line from rank 0 with whole-job units.  A functional check, not a rate."""
    line = _launch(["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--headline-only"], 2,
                   30700 + os.getpid() % 400, MMX_BENCH_SHARE_DEVICE="1", MMX_BENCH_BACKEND="gloo")
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["global_batch"] == 128
    assert abs(line["value"] - 128 / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-3


SHARED = dict(MMX_EVAL_SHARE_DEVICE="1", MMX_EVAL_BACKEND="gloo")


@pytest.mark.parametrize("script, args, same", [
    ("lxmert_perturbation_eval.py", ["--num-samples", "40", "--max-batch", "8", "--method", "ours_no_lrp"], ("samples", "step_accuracy_percent")),
    ("visualbert_pert_eval.py", ["--num-samples", "12"], ("samples", "step_accuracy_percent")),
    ("detr_masks_eval.py", ["--num-images", "6", "--graph-slots", "8", "--warmup-images", "1"], ("images", "mean_kept", "mean_mask_area")),
])
def test_evaluators_two_ranks_on_one_gpu_equal_one_rank(script, args, same):
    """The three sharded evaluators with their REAL model legs, two ranks sharing ``cuda:0`` and the one exchange step over gloo
    (``sharding.init_evaluator_process`` test hooks) against a single process: same sample list, rank-strided shards, results
    gathered back into loader order -> identical tables."""
    port = 31100 + os.getpid() % 300
    one = _launch([os.path.join("examples", script)] + args, 1, port)
    two = _launch([os.path.join("examples", script)] + args, 2, port + 301, **SHARED)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    for key in same:
        assert one[key] == two[key], (key, one[key], two[key])


def test_lxmert_evaluator_two_ranks_equal_one_rank():
    _need_two_gpus()
    args = ["examples/lxmert_perturbation_eval.py", "--num-samples", "96", "--max-batch", "16", "--method", "ours_no_lrp"]
    one = _launch(args, 1, 29900 + os.getpid() % 400)
    two = _launch(args, 2, 30300 + os.getpid() % 400)
    assert two["n_gpus"] == 2 and one["samples"] == two["samples"] == 96
    assert one["step_accuracy_percent"] == two["step_accuracy_percent"]
