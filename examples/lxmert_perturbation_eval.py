"""Sharded LXMERT perturbation evaluation on synthetic data -- the shape of ``lxmert/lxmert/perturbation.py``'s main
loop (BASELINE.json config 4) on this package: one process per GPU, samples sharded rank-strided, batches of items of
ANY question length (padded; the schedule kernel takes per-sample lengths) explained from ONE captured hipGraph and
perturbed as one batch, ONE all-gather of the per-sample step accuracies at the end.

    python examples/lxmert_perturbation_eval.py --num-samples 512                      # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \\
        examples/lxmert_perturbation_eval.py --num-samples 10000                       # one rank per GPU, RCCL

Random-init LXMERT-base and random features / questions (no VQA data or checkpoint offline): the numbers that mean
something are samples/s, not the accuracies.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from transformer_mm_explainability_amd import lxmert_explainability as le  # noqa: E402
from transformer_mm_explainability_amd import lxmert_model as lm  # noqa: E402
from transformer_mm_explainability_amd import lxmert_perturbation as lp  # noqa: E402
from transformer_mm_explainability_amd import sharding  # noqa: E402


def synthetic_item(k, regions, feat_dim, vocab, answers):
    """Item ``k`` of the synthetic dataset (seeded by its index: any rank can materialise any item)."""
    g = torch.Generator().manual_seed(1000 + k)
    T = int(torch.randint(6, 21, (1,), generator=g))
    label = torch.zeros(answers)
    label[torch.randint(0, answers, (3,), generator=g)] = torch.tensor([1.0, 0.6, 0.3])
    return dict(input_ids=torch.randint(1, vocab, (T,), generator=g), visual_feats=torch.randn(regions, feat_dim, generator=g),
                visual_pos=torch.rand(regions, 4, generator=g), label=label)


# methods of the reference's --method flag (lxmert/lxmert/perturbation.py:216-245).  The first group is explained a whole
# batch at a time (rule flags of GeneratorOurs.generate_ours_batch); the others run the reference's per-item generator call
# (lxmert_explainability.GeneratorOurs / GeneratorBaselines / GeneratorOursAblationNoAggregation; the LRP ones on the body's
# own relprop pass) followed by the same batched perturbation of that one item.
BATCHED_METHODS = {"ours_no_lrp": {}, "ours_no_lrp_no_norm": {"normalize_self_attention": False},
                   "ablation_no_self_in_10": {"apply_self_in_rule_10": False}}
OTHER_METHODS = {"ours_with_lrp", "rollout", "partial_lrp", "transformer_att", "raw_attn", "attn_gradcam",
                 "ours_with_lrp_no_normalization", "ablation_no_aggregation"}


class ItemUsage:
    """``ModelUsage`` of the reference for one item (lxmert/lxmert/perturbation.py:45-83): ``forward`` runs the body and leaves
    the item's text / region counts where the generators read them."""

    def __init__(self, model):
        self.model = model

    def forward(self, inputs):
        self.text_len, self.image_boxes_len = inputs["input_ids"].shape[1], inputs["visual_feats"].shape[1]
        return self.model(**inputs)


def per_item_method(name, le, usage):
    """The reference's dispatch (perturbation.py:218-243) -> ``item -> (R_t_t, R_t_i)``."""
    ours, base, abl = le.GeneratorOurs(usage), le.GeneratorBaselines(usage), le.GeneratorOursAblationNoAggregation(usage)
    return {"transformer_att": base.generate_transformer_attr, "attn_gradcam": base.generate_attn_gradcam,
            "partial_lrp": base.generate_partial_lrp, "raw_attn": base.generate_raw_attn, "rollout": base.generate_rollout,
            "ours_with_lrp_no_normalization": lambda it: ours.generate_ours(it, normalize_self_attention=False),
            "ours_with_lrp": lambda it: ours.generate_ours(it, use_lrp=True),
            "ablation_no_aggregation": lambda it: abl.generate_ours_no_agg(it, use_lrp=False, normalize_self_attention=False),
            }[name]


def ref_bool(text):
    """The reference declares these flags with ``type=bool`` (any non-empty string is True there, even "False"); here
    "false" / "0" / "no" / "" mean False."""
    return str(text).strip().lower() not in ("", "0", "false", "no", "off")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-samples", type=int, default=512)
    ap.add_argument("--dataset-len", type=int, default=20000)
    ap.add_argument("--max-batch", type=int, default=32)
    # the reference evaluator's own flags (lxmert/lxmert/src/param.py:93-107), same names and defaults
    ap.add_argument("--method", type=str, default="ours_no_lrp", choices=sorted(BATCHED_METHODS) + sorted(OTHER_METHODS))
    ap.add_argument("--is-positive-pert", type=ref_bool, default=False, help="positive perturbation test (default: negative)")
    ap.add_argument("--is-text-pert", type=ref_bool, default=False, help="text perturbation test (default: image)")
    ap.add_argument("--text", dest="is_text_pert", action="store_true", help="alias of --is-text-pert True")
    ap.add_argument("--positive", dest="is_positive_pert", action="store_true", help="alias of --is-positive-pert True")
    ap.add_argument("--resume-dir", default=None, help="per-rank partial score files; finished samples are skipped on restart")
    ap.add_argument("--eager-perturbation", action="store_true",
                    help="round-5 path: the 9-step image test as eager PyTorch launches instead of one hipGraph replay")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="round-5 host path: batch assembly and blocking host-to-device copies on the launching thread")
    ap.add_argument("--bucket-by-length", action="store_true",
                    help="round-1 behaviour: group items by question length, eager explain pass per group")
    args = ap.parse_args()
    args.text, args.positive = args.is_text_pert, args.is_positive_pert
    per_item = args.method not in BATCHED_METHODS
    if per_item:
        args.max_batch, args.bucket_by_length = 1, True                # one item per explain call, its own length
    rule_flags = BATCHED_METHODS.get(args.method, {})
    rank, world, dev, gather_dev = sharding.init_evaluator_process()
    if world > 1:
        import torch.distributed as dist

    cfg = lm.LxmertConfig()
    torch.manual_seed(0)
    model = lm.LxmertForQuestionAnswering(cfg).to(dev).eval()
    indices = sharding.perturbation_sample_indices(args.dataset_len, args.num_samples)      # same list on every rank
    gen = le.GeneratorOurs(type("Usage", (), {"model": model})())
    pert = lp.LxmertPerturbation(model)
    run_cfg = {"evaluator": "lxmert_perturbation", "method": args.method, "test": "text" if args.text else "image",
               "positive": bool(args.positive), "steps": list(lp.PERT_STEPS), "dataset_len": args.dataset_len}
    store = sharding.PartialScores(args.resume_dir, rank, config=run_cfg) if args.resume_dir else None
    cache = {}

    def item(k):
        if k not in cache:
            cache[k] = synthetic_item(k, 36, cfg.visual_feat_dim, cfg.vocab_size, cfg.num_qa_labels)
        return cache[k]

    T_PAD = 20                                 # the synthetic questions have 6..20 tokens
    graphed = {}

    def load_batch(ids):                       # HOST half of a batch (sharding.BatchPrefetcher runs it on a worker thread, ahead)
        items = [cache.pop(k, None) or synthetic_item(k, 36, cfg.visual_feat_dim, cfg.vocab_size, cfg.num_qa_labels) for k in ids]
        n = len(items)
        if args.bucket_by_length:
            B, T = n, items[0]["input_ids"].numel()
        else:                                  # fixed shape: pad the tail batch with copies of its last item
            B, T = args.max_batch, T_PAD
            items = items + [items[-1]] * (B - n)
        input_ids = torch.zeros(B, T, dtype=torch.long)
        mask = torch.zeros(B, T)
        for b, it in enumerate(items):
            t = it["input_ids"].numel()
            input_ids[b, :t] = it["input_ids"]
            mask[b, :t] = 1
        # lists of per-item tensors are stacked by the prefetcher straight into its pinned staging buffers
        return dict(input_ids=input_ids, attention_mask=mask, token_type_ids=torch.zeros(B, T, dtype=torch.long),
                    visual_feats=[it["visual_feats"] for it in items], visual_pos=[it["visual_pos"] for it in items],
                    labels=[it["label"] for it in items])

    def process_batch(ids, batch=None):        # explain + perturb one batch -> [len(ids), 9] accuracies
        if batch is None:                      # --no-prefetch: the round-5 path (host assembly + blocking copies on this thread)
            batch = {k: (torch.stack(v) if isinstance(v, list) else v).to(dev) for k, v in load_batch(ids).items()}
        n = len(ids)
        labels = batch.pop("labels")
        if per_item:
            if "explain" not in graphed:
                graphed["explain"] = per_item_method(args.method, le, ItemUsage(model))
            usage_item = dict(batch)
            R_t_t, R_t_i = (r.unsqueeze(0) for r in graphed["explain"](usage_item))
        elif args.bucket_by_length:
            R_t_t, R_t_i = gen.generate_ours_batch(batch, **rule_flags)
        else:
            if "run" not in graphed:           # captured once; serves every later batch whatever its question lengths
                graphed["run"] = le.GraphedGenerateOursBatch(model, batch, **rule_flags)
            R_t_t, R_t_i = graphed["run"](batch, check="deferred")      # (the diag >= 0 word is asserted one batch later: no stall)
        if not args.text and not per_item and not args.bucket_by_length and not args.eager_perturbation:
            if "pert" not in graphed:          # the image test of a fixed-shape batch: one more hipGraph (see GraphedImagePerturbation)
                graphed["pert"] = lp.GraphedImagePerturbation(pert, batch, R_t_t, R_t_i, labels, args.positive)
            return graphed["pert"](batch, R_t_t, R_t_i, labels)[:n].clone()
        cam_image, cam_text = lp.normalize_cams_batch(R_t_t, R_t_i, batch["attention_mask"])
        scores = pert.perturbation_text(batch, cam_text, args.positive) if args.text else \
            pert.perturbation_image(batch, cam_image, args.positive)
        return lp.LxmertPerturbation.accuracy(scores, labels)[:n]

    torch.cuda.synchronize()
    stats = {}
    t0 = time.perf_counter()
    length_of = (lambda k: item(k)["input_ids"].numel()) if args.bucket_by_length else (lambda k: 0)   # 0: one bucket
    per_sample = sharding.evaluate_sharded(indices, length_of, process_batch,
                                           len(lp.PERT_STEPS), max_batch=args.max_batch, store=store, device=gather_dev,
                                           load_batch=None if args.no_prefetch else load_batch, prefetch_device=dev, stats=stats)
    if "run" in graphed:
        graphed["run"].finish()                # the last batch's handle_residual word
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=gather_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    acc = sharding.mean_step_accuracy(per_sample)
    if rank == 0:
        print(json.dumps({"samples": len(indices), "n_gpus": world, "seconds": round(elapsed, 3),
                          "samples_per_s": round(len(indices) / elapsed, 1), "test": "text" if args.text else "image",
                          "method": args.method, "positive": bool(args.positive), "host_path": "inline" if args.no_prefetch else "prefetch thread + side-stream copies",
                          # the first batch holds the hipGraph capture and the library warm-up: the rate of the remaining batches
                          "samples_per_s_after_first_batch": round((len(indices) - min(args.max_batch, len(indices))) / max(elapsed - (stats.get("first_batch_s") or 0.0), 1e-9) / world, 1) * world
                          if stats.get("batches", 0) > 1 else None,
                          "first_batch_s": round(stats.get("first_batch_s") or 0.0, 3),
                          "starved_for_host_s": None if stats.get("starved_s") is None else round(stats["starved_s"], 3),
                          "step_accuracy_percent": [round(float(a), 2) for a in acc]}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
