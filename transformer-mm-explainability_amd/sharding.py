"""Sample sharding + result gathering for the perturbation / segmentation evaluators.

Every relevancy map depends on one sample only, so the evaluators shard the sample list across ranks (one process
per GPU) and exchange per-sample results ONCE with a fixed-shape all-gather (RCCL on the GPUs, gloo in the CPU tests).
Nothing here is on the per-sample data path.

Reference call sites this serves: ``lxmert/lxmert/perturbation.py:205-210`` (seeded shuffle, first ``num_samples``
items, single process), ``VisualBERT/mmf/trainers/core/evaluation_loop.py:104-168`` (running per-step accuracy),
``DETR/main.py:151-153`` (``DistributedSampler(shuffle=False)``).
"""
from __future__ import annotations

import json
import os
import random

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL between processes (before the HIP runtime starts)

import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_evaluator_process():
    """Set up one evaluator process of a one-process-per-GPU launch (``python -m torch.distributed.run --nproc-per-node N
    examples/..._eval.py``; a plain ``python`` run is world size 1): picks ``cuda:LOCAL_RANK``, joins the RCCL group when
    ``WORLD_SIZE > 1``.  Returns ``(rank, world_size, device, gather_device)``; ``gather_device`` is where the per-sample
    score table lives for the one exchange step (the GPU with RCCL).

    Test hooks for 1-GPU boxes, never set by a launcher: ``MMX_EVAL_SHARE_DEVICE=1`` puts every rank on ``cuda:0`` and
    ``MMX_EVAL_BACKEND=gloo`` swaps RCCL (which refuses two ranks per device) for gloo with the table on the host."""
    rank, world_size = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    backend = os.environ.get("MMX_EVAL_BACKEND", "nccl")
    device = torch.device("cuda", 0 if os.environ.get("MMX_EVAL_SHARE_DEVICE") else int(os.environ.get("LOCAL_RANK", 0)))
    torch.cuda.set_device(device)
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world_size)
    return rank, world_size, device, device if backend == "nccl" else torch.device("cpu")


def perturbation_sample_indices(dataset_len, num_samples, seed=1234):
    """The reference's sample selection (perturbation.py:205-210): ``random.seed(1234)``, shuffle ``range(len)``,
    keep the first ``num_samples``.  Identical on every rank."""
    rng = random.Random(seed)
    idx = list(range(dataset_len))
    rng.shuffle(idx)
    return idx[:num_samples]


def shard_indices(indices, rank=None, world_size=None):
    """Rank-strided slice: rank r owns ``indices[r::world_size]`` (deterministic, balanced to within one sample)."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    return list(indices[rank::world_size])


def length_buckets(lengths, max_batch):
    """Batches of samples with EQUAL length (the batched generators take no padding, see
    ``lxmert_explainability.GeneratorOurs.generate_ours_batch``): ``[(length, [positions...]), ...]`` over positions
    ``0 .. len(lengths) - 1``, each batch at most ``max_batch`` long, in a deterministic order (by length, then
    position)."""
    by_len = {}
    for pos, n in enumerate(lengths):
        by_len.setdefault(int(n), []).append(pos)
    out = []
    for n in sorted(by_len):
        group = by_len[n]
        for i in range(0, len(group), max_batch):
            out.append((n, group[i:i + max_batch]))
    return out


def gather_per_sample(local, total, fill=float("nan"), collective_at_world_one=False):
    """All-gather per-sample rows back into the ORIGINAL (unsharded) order.

    ``local``: ``[n_local, ...]`` results of this rank's ``shard_indices`` slice, in that order.  ``total``: number of
    samples over all ranks.  Returns ``[total, ...]`` on every rank.  One fixed-shape collective: shards are padded to
    ``ceil(total / world)`` rows with ``fill`` and the padding is dropped after the gather.
    """
    rank, world_size = world()
    if world_size == 1 and not (collective_at_world_one and dist.is_available() and dist.is_initialized()):
        return local          # (``collective_at_world_one``: the RCCL smoke test runs the padded all-gather in a world-size-1 group)
    per = (total + world_size - 1) // world_size
    pad = torch.full((per,) + tuple(local.shape[1:]), fill, dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world_size * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad.contiguous())
    out = out.view((world_size, per) + tuple(local.shape[1:]))
    # sample k of the original order lives on rank k % world at row k // world
    return out.transpose(0, 1).reshape((per * world_size,) + tuple(local.shape[1:]))[:total]


def all_gather_objects(data, device=None):
    """Every rank's picklable ``data`` as a list in rank order, on every rank -- what ``DETR/util/misc.py:88-128`` ``all_gather`` gives the
    reference's evaluators (COCO ``evalImgs`` / image-id lists of different lengths per rank, ``DETR/datasets/coco_eval.py:170-189``).  World
    size 1 returns ``[data]`` without touching a process group, like the reference.  Two fixed-shape collectives: the byte counts (one int64
    per rank), then ONE flat ``all_gather_into_tensor`` of the payloads zero-padded to the longest (RCCL wants equal shapes; the reference
    pads too).  ``device``: where the byte tensors live -- defaults to the current CUDA device under the ``nccl`` backend, the host otherwise."""
    import pickle
    rank, world_size = world()
    if world_size == 1:
        return [data]
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    payload = torch.frombuffer(bytearray(pickle.dumps(data)), dtype=torch.uint8).to(device)
    sizes = torch.empty(world_size, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(sizes, torch.tensor([payload.numel()], dtype=torch.int64, device=device))
    sizes = [int(n) for n in sizes.tolist()]
    longest = max(max(sizes), 1)
    padded = torch.zeros(longest, dtype=torch.uint8, device=device)
    padded[: payload.numel()] = payload
    flat = torch.empty(world_size * longest, dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(flat, padded)
    flat = flat.cpu().numpy()
    return [pickle.loads(flat[r * longest: r * longest + sizes[r]].tobytes()) for r in range(world_size)]


def merge_eval_images(img_ids, eval_imgs, device=None):
    """``DETR/datasets/coco_eval.py:170-189`` ``merge``: gather every rank's image ids (a list) and its per-image evaluation array
    (``[..., n_local_images]``, images on the LAST axis), concatenate in rank order and keep each image ONCE, in sorted id order (a
    distributed sampler pads the last shard with repeated images: the first occurrence wins, as ``np.unique(return_index=True)`` picks it).
    Returns ``(merged_img_ids, merged_eval_imgs)`` as numpy arrays, identical on every rank."""
    import numpy as np
    all_ids = all_gather_objects(list(img_ids), device)
    all_imgs = all_gather_objects(np.asarray(eval_imgs), device)
    merged_ids = np.array([i for part in all_ids for i in part])
    merged_imgs = np.concatenate(all_imgs, axis=-1)
    merged_ids, first = np.unique(merged_ids, return_index=True)
    return merged_ids, merged_imgs[..., first]


class BatchPrefetcher:
    """Host side of an evaluator loop off the critical path: a worker thread assembles batch i + 1, i + 2 (``load(ids)`` -> a dict of
    CPU tensors: dataset access, padding, stacking), stages them in REUSED pinned buffers and issues the host-to-device copies on a
    side stream, while the main thread launches the device work of batch i.  Iterating yields ``(ids, batch_on_device)``; the
    consumer's stream waits on the copy's event (no host synchronisation anywhere).

    Why: the reference loop (lxmert/lxmert/perturbation.py:205-251) and round 5's port of it built every batch on the thread that
    also launches the kernels, from pageable memory with blocking copies -- 465-507 samples / s end to end against 1170 samples / s
    for the device work alone, i.e. with one process per GPU the host, not the GPUs or xGMI, set the rate."""

    def __init__(self, batches, load, device, depth=2, loaders=2):
        """``loaders``: threads that run ``load`` (the host assembly of a batch) for the batches ahead; ONE staging thread takes their
        results in order, copies them into the pinned sets and starts the device copies.  With a launching thread that no longer
        stalls per batch, a single loader thread sharing the interpreter with it was the bottleneck (round 6: ``starved_s``)."""
        import queue
        import threading
        self.device = torch.device(device)
        self.load = load
        self.loaders = max(1, int(loaders))
        self.batches = list(batches)
        self.q = queue.Queue(maxsize=depth)
        self.cuda = self.device.type == "cuda"
        self.stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.pool, self.turn, self.sets = {}, 0, depth + 2       # pinned staging sets: one per batch that can be in flight + 1
        self.error = None
        self.thread = threading.Thread(target=self._work, daemon=True)
        self.thread.start()

    def _pinned(self, name, t):
        """``t``: a tensor, or a LIST of equally shaped tensors that is stacked straight into the staging buffer (no intermediate
        ``torch.stack`` result: a fresh 9 MB host allocation per batch costs more in page faults than the copy itself)."""
        many = isinstance(t, (list, tuple))
        shape = ((len(t),) + tuple(t[0].shape)) if many else tuple(t.shape)
        dtype = t[0].dtype if many else t.dtype
        key = (name, shape, dtype, self.turn % self.sets)
        buf = self.pool.get(key)
        if buf is None:
            buf = self.pool[key] = torch.empty(shape, dtype=dtype).pin_memory() if self.cuda else torch.empty(shape, dtype=dtype)
        if many:
            for row, part in zip(buf, t):
                row.copy_(part)
        else:
            buf.copy_(t)
        return buf

    @staticmethod
    def _is_data(v):
        return torch.is_tensor(v) or (isinstance(v, (list, tuple)) and len(v) > 0 and all(torch.is_tensor(x) for x in v))

    def _work(self):
        import collections
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=self.loaders)
        try:
            ahead, todo = collections.deque(), iter(self.batches)

            def submit():
                ids = next(todo, None)
                if ids is not None:
                    ahead.append((ids, pool.submit(self.load, ids)))
            for _ in range(self.loaders + 1):
                submit()
            while ahead:
                ids, loaded = ahead.popleft()
                host = loaded.result()
                submit()
                if self.cuda:
                    with torch.cuda.stream(self.stream):
                        dev = {k: (self._pinned(k, v).to(self.device, non_blocking=True) if self._is_data(v) else v)
                               for k, v in host.items()}
                        done = torch.cuda.Event()
                        done.record(self.stream)
                else:
                    dev, done = {k: (self._pinned(k, v).clone() if self._is_data(v) else v) for k, v in host.items()}, None
                self.turn += 1
                self.q.put((ids, dev, done))
        except BaseException as exc:                      # surfaced on the consumer's thread
            self.error = exc
        finally:
            pool.shutdown(wait=False, cancel_futures=True)
        self.q.put(None)

    def __iter__(self):
        import time
        self.starved_s = 0.0                                  # time the consumer spent waiting for the worker (host-bound if large)
        while True:
            t0 = time.perf_counter()
            item = self.q.get()
            self.starved_s += time.perf_counter() - t0
            if item is None:
                if self.error is not None:
                    raise self.error
                return
            ids, dev, done = item
            if done is not None:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(done)
                for v in dev.values():
                    if torch.is_tensor(v):
                        v.record_stream(cur)              # allocated on the side stream, consumed on this one
            yield ids, dev


def evaluate_sharded(sample_ids, length_of, process_batch, n_cols, max_batch=128, store=None, device="cpu", load_batch=None,
                     prefetch_device=None, stats=None):
    """The sharded evaluator loop in one place (``examples/lxmert_perturbation_eval.py`` is this plus a model).

    ``sample_ids``: the FULL, identically ordered sample list (every rank passes the same one).  This rank takes its
    rank-strided shard, skips what ``store`` (a ``PartialScores``) already holds, buckets the rest by ``length_of(id)``
    (equal-length batches of at most ``max_batch``), calls ``process_batch(ids) -> [len(ids), n_cols]`` per bucket,
    records the rows in ``store`` if given, and finally all-gathers every rank's rows back into ``sample_ids`` order.
    Returns ``[len(sample_ids), n_cols]`` on every rank.  The only collective is that final gather.

    ``load_batch(ids) -> dict of CPU tensors`` (optional): the host half of a batch, run by a ``BatchPrefetcher`` worker thread two
    batches ahead with the copies to ``prefetch_device`` on a side stream; ``process_batch`` is then called as
    ``process_batch(ids, batch_on_device)``.  With a ``store`` the rows of batch i are written while batch i + 1 runs (one batch of
    lag: the device-to-host read of a batch's rows does not stall the launch of the next).  ``stats`` (a dict, optional) receives
    ``batches``, ``first_batch_s`` (captures / warm-up happen there) and ``starved_s`` (time this thread waited for the loader).
    """
    mine = shard_indices(sample_ids)
    done = store.done() if store is not None else ()           # computed once (a 10k-sample resume: not once per id)
    todo = [k for k in mine if k not in done]
    fresh = {}
    id_batches = [[todo[p] for p in positions] for _, positions in length_buckets([length_of(k) for k in todo], max_batch)]
    import time
    feeder = BatchPrefetcher(id_batches, load_batch, prefetch_device or device) if load_batch is not None else None
    pending = None                                              # (ids, rows) of the previous batch, not yet in the store
    t_start, first_s = time.perf_counter(), None
    for ids, prepared in (feeder if feeder is not None else ((ids, None) for ids in id_batches)):
        rows = process_batch(ids, prepared) if load_batch is not None else process_batch(ids)
        if first_s is None:
            first_s = time.perf_counter() - t_start
        if tuple(rows.shape) != (len(ids), n_cols):
            raise ValueError("process_batch returned %s for %d ids x %d columns" % (tuple(rows.shape), len(ids), n_cols))
        if store is not None:
            if load_batch is None:
                store.add(ids, rows)
            else:                                               # write the PREVIOUS batch: its rows are ready, this one's are in flight
                if pending is not None:
                    store.add(pending[0], _landed(pending[1]))
                pending = (ids, _to_host_async(rows))
        else:
            for k, row in zip(ids, rows):
                fresh[k] = row
    if pending is not None:
        store.add(pending[0], _landed(pending[1]))
    if stats is not None:
        stats.update(batches=len(id_batches), first_batch_s=first_s, starved_s=getattr(feeder, "starved_s", None))
    if store is not None and mine:
        local = store.table(mine, device=device)
    elif mine:
        local = torch.stack([fresh[k] for k in mine]).to(device=device, dtype=torch.float32)
    else:
        local = torch.zeros(0, n_cols, dtype=torch.float32, device=device)
    return gather_per_sample(local, len(sample_ids))


_D2H_STREAMS = {}


def _to_host_async(rows):
    """Start the device -> host copy of a batch's result rows on a side stream (pinned destination, behind an event recorded where
    the rows were produced) -> ``(host rows, event)``; a blocking ``.cpu()`` on the launching stream would be queued behind the NEXT
    batch's launches and stall the thread for a whole batch.  CPU rows pass through."""
    if not rows.is_cuda:
        return rows, None
    ready = torch.cuda.Event()
    ready.record()
    side = _D2H_STREAMS.setdefault(rows.device, torch.cuda.Stream(rows.device))
    side.wait_event(ready)
    host = torch.empty(rows.shape, dtype=rows.dtype).pin_memory()
    with torch.cuda.stream(side):
        host.copy_(rows, non_blocking=True)
        done = torch.cuda.Event()
        done.record(side)
    rows.record_stream(side)
    return host, done


def _landed(pair):
    host, done = pair
    if done is not None:
        done.synchronize()
    return host


def mean_step_accuracy(per_sample_scores):
    """``[total, n_steps]`` per-sample scores -> the evaluators' printed metric: mean over samples x 100
    (perturbation.py:250-251).  NaN padding rows are never present after ``gather_per_sample``."""
    return per_sample_scores.double().mean(dim=0) * 100.0


class PartialScores:
    """Append-only per-rank store of per-sample result rows, so that a killed evaluator resumes where it stopped.

    The reference evaluators keep their running accuracies in Python lists (``perturbation.py:43``,
    ``evaluation_loop.py:97``): a crash at sample 9 000 of 10 000 loses everything.  Here every finished batch is
    appended as one JSON line ``{"ids": [...], "rows": [[...], ...]}`` to ``<directory>/scores_rank<r>.jsonl`` (flushed
    and fsynced); on restart ``done()`` tells which sample ids can be skipped and ``table()`` returns the rows in the
    order the gather expects.  A torn last line (killed mid-write) is ignored.
    """

    def __init__(self, directory, rank, config=None):
        """``config``: any JSON-serialisable description of WHAT is being scored (method, test type, steps ...).  It is
        written as the file's header record and must match on resume: rows of another method / test are never reused."""
        os.makedirs(directory, exist_ok=True)
        self.path = os.path.join(directory, "scores_rank%d.jsonl" % rank)
        self.rows = {}
        header = None
        if os.path.exists(self.path):
            with open(self.path) as f:
                for line in f:
                    try:
                        rec = json.loads(line)
                    except ValueError:
                        continue                      # torn tail of an interrupted write
                    if "config" in rec:
                        header = rec["config"]
                        continue
                    for k, row in zip(rec["ids"], rec["rows"]):
                        self.rows[int(k)] = row
        config = json.loads(json.dumps(config)) if config is not None else None       # normalised (tuples -> lists)
        if (self.rows or header is not None) and header != config:
            raise ValueError("%s holds rows of another run (%r), not of %r: use a fresh directory" %
                             (self.path, header, config))
        # header wanted whenever the file has neither a header nor rows yet -- also when all it holds is the torn first
        # line of a run killed during its very first write (non-empty file, nothing parsed)
        fresh = header is None and not self.rows
        torn = os.path.exists(self.path) and os.path.getsize(self.path) > 0 and \
            open(self.path, "rb").read()[-1:] != b"\n"
        self._f = open(self.path, "a")
        if torn:
            self._f.write("\n")                  # finish the torn line so the next record starts on its own line
        if fresh and config is not None:
            self._f.write(json.dumps({"config": config}) + "\n")
            self._f.flush()

    def done(self):
        return set(self.rows)

    def add(self, sample_ids, rows):
        """``sample_ids``: iterable of ints; ``rows``: ``[len(ids), n_cols]`` tensor (moved to the host here)."""
        ids = [int(k) for k in sample_ids]
        data = rows.detach().cpu().tolist()
        for k, row in zip(ids, data):
            self.rows[k] = row
        self._f.write(json.dumps({"ids": ids, "rows": data}) + "\n")
        self._f.flush()
        os.fsync(self._f.fileno())

    def table(self, sample_ids, device="cpu"):
        """Rows of ``sample_ids`` in that order (every id must be done) -> ``[len(ids), n_cols]`` fp32 tensor."""
        rows = [self.rows[int(k)] for k in sample_ids]
        if not rows:
            raise ValueError("PartialScores.table: empty id list (the column count is unknown here)")
        return torch.tensor(rows, dtype=torch.float32, device=device)

    def close(self):
        self._f.close()
