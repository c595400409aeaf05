"""Sharded VisualBERT perturbation evaluation on synthetic features -- the shape of
``VisualBERT/mmf/trainers/core/evaluation_loop.py:73-169`` (``run.py --trainer=mmf_pert``) on this package: one process
per GPU, samples sharded rank-strided, the generator's relevancy row per item, the 9 perturbed re-runs of an item in ONE
forward, ONE all-gather of the per-sample step accuracies at the end.

    python examples/visualbert_pert_eval.py --num-samples 512                         # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29513 \\
        examples/visualbert_pert_eval.py --num-samples 10000 --text                   # one rank per GPU, RCCL

``--reference-exact``: the reference loop stops only AFTER item ``num_samples + 1`` (its test is ``i > num_samples``) and
still divides the accumulated accuracies by ``num_samples``; with the flag the printed numbers use
exactly that accounting, without it ``num_samples`` items are evaluated and averaged.
Random-init BERT-base VisualBERT and random features / questions: the numbers that mean something are samples/s.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from transformer_mm_explainability_amd import sharding  # noqa: E402

N_STEPS = 9


def synthetic_item(k, vocab=30522, regions=100, feat_dim=2048, labels=3129, pad_to=24):
    """Item ``k`` of the synthetic dataset (seeded by its index): padded question, region features, soft VQA targets."""
    g = torch.Generator().manual_seed(9000 + k)
    n_text = int(torch.randint(8, 21, (1,), generator=g))
    ids = torch.zeros(1, pad_to, dtype=torch.long)
    ids[0, :n_text] = torch.randint(1, vocab, (n_text,), generator=g)
    mask = torch.zeros(1, pad_to, dtype=torch.long)
    mask[0, :n_text] = 1
    targets = torch.zeros(1, labels)
    targets[0, torch.randint(0, labels, (3,), generator=g)] = torch.tensor([1.0, 0.6, 0.3])
    return {"input_ids": ids, "input_mask": mask, "segment_ids": torch.zeros(1, pad_to, dtype=torch.long),
            "image_feature_0": torch.randn(1, regions, feat_dim, generator=g), "targets": targets}


def evaluate(sample_ids, step_accuracies_of, num_samples, reference_exact=False, store=None, device="cpu"):
    """Sharded ``evaluation_loop``: ``step_accuracies_of(id) -> [9]`` (the item's ``targets[argmax]`` after each of the 9
    perturbation steps).  ``sample_ids``: the loader order; the reference's accounting evaluates ``num_samples + 1`` of
    them.  Returns ``(per_sample [n, 9], printed [9])`` on every rank; ``printed`` = sum / num_samples * 100."""
    n = min(len(sample_ids), num_samples + 1 if reference_exact else num_samples)
    ids = list(sample_ids[:n])
    table = sharding.evaluate_sharded(ids, lambda k: 0, lambda batch: torch.stack([step_accuracies_of(k) for k in batch]),
                                      N_STEPS, max_batch=1, store=store, device=device)
    return table, table.double().sum(dim=0) / num_samples * 100.0


def ref_bool(text):
    """The reference declares these flags with ``type=bool`` (any non-empty string is True there, even "False"); here
    "false" / "0" / "no" / "" mean False."""
    return str(text).strip().lower() not in ("", "0", "false", "no", "off")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-samples", type=int, default=256)
    ap.add_argument("--dataset-len", type=int, default=20000)
    # the reference's own flags (VisualBERT/mmf/utils/flags.py:29-46), same names, choices and defaults
    ap.add_argument("--method", type=str, default="ours_no_lrp",
                    choices=["ours_no_lrp", "transformer_attribution", "partial_lrp", "raw_attn", "attn_gradcam", "rollout"])
    ap.add_argument("--is-positive-pert", type=ref_bool, default=False)
    ap.add_argument("--is-text-pert", type=ref_bool, default=False)
    ap.add_argument("--text", dest="is_text_pert", action="store_true", help="alias of --is-text-pert True")
    ap.add_argument("--positive", dest="is_positive_pert", action="store_true", help="alias of --is-positive-pert True")
    ap.add_argument("--reference-exact", action="store_true")
    ap.add_argument("--resume-dir", default=None)
    args = ap.parse_args()
    args.text, args.positive = args.is_text_pert, args.is_positive_pert
    rank, world, dev, gather_dev = sharding.init_evaluator_process()
    if world > 1:
        import torch.distributed as dist
    from transformer_mm_explainability_amd import visualbert_explainability as vb
    from transformer_mm_explainability_amd import visualbert_model as vm
    from transformer_mm_explainability_amd import visualbert_perturbation as vp

    torch.manual_seed(0)
    model = vm.VisualBERT(vm.VisualBertConfig()).to(dev).eval()
    gen = vb.SelfAttentionGenerator(model)
    pert = vp.VisualBertPerturbation(model)

    # evaluation_loop.py:82-87: the method table of the reference (the LRP ones need a body with relprop and raise otherwise)
    method_expl = {"transformer_attribution": gen.generate_transformer_att, "ours_no_lrp": gen.generate_ours,
                   "partial_lrp": gen.generate_partial_lrp, "raw_attn": gen.generate_raw_attn,
                   "attn_gradcam": gen.generate_attn_gradcam, "rollout": gen.generate_rollout}[args.method]

    def step_accuracies_of(k):
        item = {name: t.to(dev) for name, t in synthetic_item(k).items()}
        cam = method_expl(dict(item)).detach()
        run = pert.perturbation_text if args.text else pert.perturbation_image
        return pert.accuracy(run(item, cam, args.positive), item["targets"][0])

    cfg = {"evaluator": "visualbert_pert", "method": args.method, "test": "text" if args.text else "image",
           "positive": bool(args.positive), "reference_exact": bool(args.reference_exact)}
    store = sharding.PartialScores(args.resume_dir, rank, config=cfg) if args.resume_dir else None
    ids = list(range(args.dataset_len))         # the loader order (mmf's sampler is sequential for evaluation)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    table, printed = evaluate(ids, step_accuracies_of, args.num_samples, args.reference_exact, store=store, device=gather_dev)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=gather_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({"samples": table.shape[0], "n_gpus": world, "seconds": round(elapsed, 3),
                          "samples_per_s": round(table.shape[0] / elapsed, 1), "test": "text" if args.text else "image",
                          "reference_exact": bool(args.reference_exact),
                          "step_accuracy_percent": [round(float(a), 2) for a in printed]}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
