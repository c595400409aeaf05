"""CPU, world_size = 2, gloo: the sharded loops of the DETR ``--masks`` and VisualBERT ``mmf_pert`` evaluator drivers
(``examples/detr_masks_eval.py``, ``examples/visualbert_pert_eval.py``) reassemble per-image / per-sample rows in the
loader order and reproduce the single-process result; the model legs are stand-ins (the attention op has no CPU path --
the real legs are covered by the ``-m gpu`` tests of ``MaskGenerator`` / ``VisualBertPerturbation``)."""
import importlib.util
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "examples", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _fake_masks(k):
    """Deterministic stand-in of ``MaskGenerator.get_masks``: image k keeps (k % 5) + 1 queries with striped masks."""
    Q, h, w = 12, 4, 6
    keep = torch.zeros(Q, dtype=torch.bool)
    keep[: (k % 5) + 1] = True
    masks = torch.full((1, Q, h, w), -1.0)
    for q in range(int(keep.sum())):
        masks[0, q] = ((torch.arange(h * w).reshape(h, w) + q + k) % 3 == 0).float() * 255
    return masks, keep


def _fake_steps(k):
    return torch.tensor([((k * 7 + s * 3) % 10) / 10.0 for s in range(9)])


def _worker(rank, world_size, port, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    detr, vbert = _load("detr_masks_eval"), _load("visualbert_pert_eval")
    ids = list(range(11))
    table = detr.evaluate(ids, _fake_masks)
    want = torch.stack([detr.image_stats(*_fake_masks(k), k) for k in ids])
    ok_detr = torch.equal(table, want) and table[:, 3].tolist() == [float(k) for k in ids]
    out = {"detr": bool(ok_detr)}
    for exact in (False, True):
        per_sample, printed = vbert.evaluate(list(range(40)), _fake_steps, num_samples=6, reference_exact=exact)
        n = 7 if exact else 6                                    # the reference's i > num_samples lets one more item in
        rows = torch.stack([_fake_steps(k) for k in range(n)])
        out["vb_%d" % exact] = bool(torch.equal(per_sample, rows) and
                                    torch.allclose(printed, rows.double().sum(0) / 6 * 100))
    torch.save(out, os.path.join(tmpdir, "ev%d.pt" % rank))
    dist.destroy_process_group()


def test_evaluator_drivers_world2(tmp_path):
    port = 35500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in (0, 1):
        res = torch.load(tmp_path / ("ev%d.pt" % r))
        assert res == {"detr": True, "vb_0": True, "vb_1": True}, res


def test_evaluator_drivers_single_process():
    """world_size 1 (no process group): same functions, no collective."""
    detr, vbert = _load("detr_masks_eval"), _load("visualbert_pert_eval")
    table = detr.evaluate(list(range(4)), _fake_masks)
    assert table.shape == (4, detr.STAT_COLS) and table[2, 0] == 3
    per_sample, printed = vbert.evaluate(list(range(10)), _fake_steps, num_samples=3, reference_exact=True)
    assert per_sample.shape == (4, 9) and printed.shape == (9,)
