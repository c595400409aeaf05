"""CPU, world_size = 2, gloo: the evaluators' sharding + all-gather reassembles per-sample results in order."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world_size, port, total, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from transformer_mm_explainability_amd import sharding
    indices = sharding.perturbation_sample_indices(100, total, seed=1234)
    mine = sharding.shard_indices(indices)
    # a fake "per-sample 9-step score" that encodes the sample id, so ordering errors are visible
    local = torch.tensor([[float(i) + 0.1 * s for s in range(9)] for i in mine], dtype=torch.float32).reshape(len(mine), 9)
    full = sharding.gather_per_sample(local, total)
    want = torch.tensor([[float(i) + 0.1 * s for s in range(9)] for i in indices], dtype=torch.float32)
    ok = torch.equal(full, want) and not torch.isnan(full).any()
    acc = sharding.mean_step_accuracy(full)
    torch.save({"ok": bool(ok), "acc": acc, "n_local": len(mine)}, os.path.join(tmpdir, "r%d.pt" % rank))
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [10, 7, 1])
def test_shard_and_gather_world2(tmp_path, total):
    port = 29500 + (os.getpid() + total) % 2000
    mp.spawn(_worker, args=(2, port, total, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert r0["ok"] and r1["ok"]
    assert r0["n_local"] + r1["n_local"] == total and abs(r0["n_local"] - r1["n_local"]) <= 1
    assert torch.equal(r0["acc"], r1["acc"])


def test_reference_sample_selection_is_reproduced():
    """perturbation.py:205-210 uses the global ``random`` module with seed 1234."""
    import random
    from transformer_mm_explainability_amd import sharding
    random.seed(1234)
    ref = list(range(50))
    random.shuffle(ref)
    assert sharding.perturbation_sample_indices(50, 20) == ref[:20]
    assert sharding.shard_indices(list(range(10)), 1, 4) == [1, 5, 9]


def _eval_worker(rank, world_size, port, total, tmpdir, resume):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from transformer_mm_explainability_amd import sharding
    ids = sharding.perturbation_sample_indices(200, total)
    calls = []

    def process(batch_ids):                       # fake scorer: row = [id, id^2 mod 97, bucket length]
        calls.append(list(batch_ids))
        assert len({k % 5 for k in batch_ids}) == 1, "a batch must hold samples of one length"
        return torch.tensor([[float(k), float(k * k % 97), float(6 + k % 5)] for k in batch_ids])

    store = sharding.PartialScores(os.path.join(tmpdir, "partial"), rank) if resume else None
    full = sharding.evaluate_sharded(ids, lambda k: 6 + k % 5, process, 3, max_batch=4, store=store)
    want = torch.tensor([[float(k), float(k * k % 97), float(6 + k % 5)] for k in ids])
    torch.save({"ok": bool(torch.equal(full, want)), "n_calls": len(calls), "max_batch": max((len(c) for c in calls), default=0)},
               os.path.join(tmpdir, "e%d.pt" % rank))
    dist.destroy_process_group()


def test_evaluate_sharded_world2_and_resume(tmp_path):
    """Whole evaluator data flow on 2 ranks: shards -> length buckets -> batches <= max_batch -> one gather; a second
    run over the same partial-score directory recomputes nothing and returns the same table."""
    port = 31500 + os.getpid() % 2000
    mp.spawn(_eval_worker, args=(2, port, 23, str(tmp_path), True), nprocs=2, join=True)
    first = [torch.load(tmp_path / ("e%d.pt" % r)) for r in (0, 1)]
    assert all(f["ok"] for f in first) and all(0 < f["max_batch"] <= 4 for f in first)
    mp.spawn(_eval_worker, args=(2, port + 1, 23, str(tmp_path), True), nprocs=2, join=True)
    second = [torch.load(tmp_path / ("e%d.pt" % r)) for r in (0, 1)]
    assert all(s["ok"] for s in second) and all(s["n_calls"] == 0 for s in second)
    (tmp_path / "fresh").mkdir()
    mp.spawn(_eval_worker, args=(2, port + 2, 5, str(tmp_path / "fresh"), False), nprocs=2, join=True)
    assert all(torch.load(tmp_path / "fresh" / ("e%d.pt" % r))["ok"] for r in (0, 1))


def test_empty_shard_with_store_world2(tmp_path):
    """more ranks than samples -- the rank with an EMPTY shard and a resumable store must still hand a
    ``[0, n_cols]`` table to the fixed-shape gather (it used to pass shape ``[0]`` and mismatch the other rank)."""
    port = 33500 + os.getpid() % 2000
    mp.spawn(_eval_worker, args=(2, port, 1, str(tmp_path), True), nprocs=2, join=True)
    assert all(torch.load(tmp_path / ("e%d.pt" % r))["ok"] for r in (0, 1))


def test_partial_scores_header_guards_against_stale_rows(tmp_path):
    """a store written for one (method, test) must not be resumed for another."""
    sys.path.insert(0, ROOT)
    from transformer_mm_explainability_amd import sharding
    cfg = {"method": "ours_no_lrp", "test": "image", "positive": False, "steps": (0, 0.5, 1)}
    st = sharding.PartialScores(str(tmp_path), 0, config=cfg)
    st.add([3, 4], torch.tensor([[1.0, 2.0], [3.0, 4.0]]))
    st.close()
    again = sharding.PartialScores(str(tmp_path), 0, config=cfg)            # same run description: rows are reused
    assert again.done() == {3, 4} and torch.equal(again.table([4, 3]), torch.tensor([[3.0, 4.0], [1.0, 2.0]]))
    again.close()
    with pytest.raises(ValueError, match="another run"):
        sharding.PartialScores(str(tmp_path), 0, config=dict(cfg, method="rollout"))
    with pytest.raises(ValueError, match="another run"):
        sharding.PartialScores(str(tmp_path), 0)                             # header present, caller states none
    with pytest.raises(ValueError, match="empty id list"):
        sharding.PartialScores(str(tmp_path / "x"), 0).table([])


def test_partial_scores_torn_first_line_still_gets_a_header(tmp_path):
    """A run killed during its very FIRST write leaves a non-empty file with nothing parseable in it: the next start must
    still write the config header (it used to key on the file size), or the resume after that refuses its own rows."""
    import torch
    from transformer_mm_explainability_amd.sharding import PartialScores
    cfg = {"method": "ours_no_lrp", "steps": 9}
    path = tmp_path / "scores_rank0.jsonl"
    path.write_text('{"config": {"method": "ours_no')             # torn header, no newline
    ps = PartialScores(str(tmp_path), 0, cfg)
    ps.add([3, 4], torch.ones(2, 9))
    ps.close()
    again = PartialScores(str(tmp_path), 0, cfg)                   # legitimate resume of the same run
    assert again.done() == {3, 4}
    again.close()
    with pytest.raises(ValueError):
        PartialScores(str(tmp_path), 0, {"method": "rollout", "steps": 9})


def test_batch_prefetcher_feeds_evaluate_sharded_in_order(tmp_path):
    """``load_batch`` path of ``evaluate_sharded`` (the evaluators' host pipeline: a worker thread assembles batches ahead, the main
    thread consumes them): same table as the inline path, every batch's prepared tensors belong to ITS ids, the store lags one batch
    and ends complete, and an exception in the worker surfaces on the caller."""
    sys.path.insert(0, ROOT)
    from transformer_mm_explainability_amd import sharding
    ids = sharding.perturbation_sample_indices(100, 37)
    loaded = []

    def load(batch_ids):
        loaded.append(list(batch_ids))
        return {"x": torch.tensor([[float(k), float(k % 7)] for k in batch_ids]), "tag": "host-object"}

    def process(batch_ids, batch):
        assert batch["tag"] == "host-object" and batch["x"][:, 0].tolist() == [float(k) for k in batch_ids]
        return torch.cat([batch["x"], batch["x"].sum(1, keepdim=True)], dim=1)

    want = torch.tensor([[float(k), float(k % 7), float(k + k % 7)] for k in ids])
    got = sharding.evaluate_sharded(ids, lambda k: k % 3, process, 3, max_batch=5, load_batch=load, prefetch_device="cpu")
    assert torch.equal(got, want) and sorted(k for b in loaded for k in b) == sorted(ids)
    store = sharding.PartialScores(str(tmp_path), 0)
    got = sharding.evaluate_sharded(ids, lambda k: k % 3, process, 3, max_batch=5, store=store, load_batch=load, prefetch_device="cpu")
    assert torch.equal(got, want) and store.done() == set(ids)

    # several loader threads finishing OUT of order (the later batch's load returns first): batches still arrive in list order
    import time
    batches = [[i, i + 100] for i in range(12)]

    def jittery(batch_ids):
        time.sleep(0.02 if batch_ids[0] % 3 == 0 else 0.001)
        return {"x": torch.tensor([float(k) for k in batch_ids])}
    seen = [(b, d["x"].tolist()) for b, d in sharding.BatchPrefetcher(batches, jittery, "cpu", depth=2, loaders=3)]
    assert [b for b, _ in seen] == batches and all(x == [float(k) for k in b] for b, x in seen)

    def broken(batch_ids):
        raise RuntimeError("dataset went away")
    with pytest.raises(RuntimeError, match="dataset went away"):
        sharding.evaluate_sharded(ids, lambda k: 0, process, 3, max_batch=5, load_batch=broken, prefetch_device="cpu")


def _merge_worker(rank, world_size, port, tmpdir):
    sys.path.insert(0, ROOT)
    import numpy as np
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from transformer_mm_explainability_amd import sharding
    gold = np.load(os.path.join(ROOT, "tests", "golden", "detr_eval_merge.npz"))
    # ragged objects first: rank 0 a long record, rank 1 an empty one
    mine = {"rank": rank, "ids": list(range(1000)) if rank == 0 else [], "note": "x" * (3 if rank else 70000)}
    everyone = sharding.all_gather_objects(mine)
    ok = [e["rank"] for e in everyone] == [0, 1] and len(everyone[0]["ids"]) == 1000 and everyone[1]["ids"] == [] \
        and len(everyone[0]["note"]) == 70000 and everyone[1]["note"] == "xxx"
    ids, evals = sharding.merge_eval_images(list(gold["ids_rank%d" % rank]), gold["evals_rank%d" % rank])
    ok = ok and np.array_equal(ids, gold["merged_ids"]) and ids.dtype == gold["merged_ids"].dtype \
        and np.array_equal(evals, gold["merged_evals"])
    torch.save({"ok": bool(ok)}, os.path.join(tmpdir, "m%d.pt" % rank))
    dist.destroy_process_group()


def test_object_gather_and_eval_image_merge_world2(tmp_path):
    """``sharding.all_gather_objects`` / ``merge_eval_images`` = ``DETR/util/misc.py:88-128`` ``all_gather`` and
    ``DETR/datasets/coco_eval.py:170-189`` ``merge``: two gloo ranks with ragged pieces reproduce, on BOTH ranks, what the reference's own
    ``merge`` source returned for the same pieces (``tests/golden/detr_eval_merge.npz``, made by ``make_golden.py gen_detr_eval_merge``);
    without a process group the gather is ``[data]`` and the merge just sorts and de-duplicates."""
    import numpy as np
    from transformer_mm_explainability_amd import sharding
    port = 29500 + (os.getpid() + 911) % 2000
    mp.spawn(_merge_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert torch.load(tmp_path / "m0.pt")["ok"] and torch.load(tmp_path / "m1.pt")["ok"]
    assert sharding.all_gather_objects({"a": 1}) == [{"a": 1}]
    ids, evals = sharding.merge_eval_images([9, 2, 9, 4], np.arange(8.0).reshape(1, 2, 4))
    assert ids.tolist() == [2, 4, 9] and evals[0, :, :].tolist() == [[1.0, 3.0, 0.0], [5.0, 7.0, 4.0]]
