"""Sharded DETR ``--masks`` evaluation on synthetic features -- the shape of ``DETR/engine.py:153-215`` (``evaluate``: one
``MaskGenerator.get_panoptic`` per validation image, results gathered over the ranks) and ``DETR/main.py:151-153``
(``DistributedSampler(dataset_val, shuffle=False)``) on this package: one process per GPU, images sharded rank-strided,
every kept query of an image explained in ONE batched pass (replayed from a hipGraph), all Otsu masks of an image in one
launch, ONE all-gather of fixed-shape per-image mask statistics at the end.

    python examples/detr_masks_eval.py --num-images 64                               # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29512 \\
        examples/detr_masks_eval.py --num-images 5000                                # one rank per GPU, RCCL

What is NOT here (control plane around the path): the ResNet backbone (the body starts at its
feature map), COCO loading, the Hungarian criterion, pycocotools' segmentation AP.  The per-image row that is gathered
instead -- kept queries, mean mask area, a checksum of the masks -- is what a COCO evaluator would consume per image; the
numbers that mean something with random weights are images/s and queries/s.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from transformer_mm_explainability_amd import sharding  # noqa: E402

STAT_COLS = 4      # kept queries, mean mask area fraction over the kept queries, mask checksum, image id


def synthetic_features(k, channels=2048, h=25, w=38, device="cpu"):
    """Backbone feature map of image ``k`` (seeded by its index: any rank can materialise any image).  Generated on the
    device it is used on: 1.9 M host-side normal variates per image cost more than the whole explainability pass."""
    g = torch.Generator(device=device).manual_seed(5000 + k)
    return torch.randn(1, channels, h, w, generator=g, device=device) * 0.5


def image_stats(masks, keep, image_id):
    """``masks [1, Q, h, w]`` (0 / 255 on kept queries, -1 elsewhere), ``keep [Q]`` -> one fixed-shape row, computed where the
    masks live (no device -> host read: the rows of a rank are gathered once, at the end)."""
    on = (masks[0] == 255).to(torch.float32) * keep.view(-1, 1, 1)          # [Q, h, w], zero on the queries not kept
    n = keep.sum().to(torch.float32)
    total = on.sum()
    area = total / (n * on.shape[1] * on.shape[2]).clamp_min(1.0)
    return torch.stack((n, area, total % 65521, torch.as_tensor(float(image_id), device=masks.device)))


def evaluate(image_ids, masks_of, store=None, device="cpu"):
    """The sharded loop: every rank explains its rank-strided share of ``image_ids`` with ``masks_of(id) -> (masks, keep)``
    and all ranks end with the ``[len(image_ids), STAT_COLS]`` table in the ORIGINAL order.  One collective."""
    def process_batch(ids):                     # one image per "batch": images differ in their number of kept queries
        return torch.stack([image_stats(*masks_of(k), k) for k in ids])
    return sharding.evaluate_sharded(image_ids, lambda k: 0, process_batch, STAT_COLS, max_batch=1, store=store, device=device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-images", type=int, default=64)
    # the reference's flag (DETR/main.py:102-107), same name, choices and default
    ap.add_argument("--method", type=str, default="ours_no_lrp",
                    choices=["ours_with_lrp", "rollout", "partial_lrp", "transformer_att", "raw_attn", "attn_gradcam",
                             "ours_no_normalization", "ours_no_lrp", "ours_no_lrp_no_norm", "ablation_no_aggregation",
                             "ablation_no_self_in_10"])
    ap.add_argument("--graph-slots", type=int, default=16, help="target slots of the captured explain pass (0: eager)")
    ap.add_argument("--keep-top", type=int, default=8, help="random-init logits are flat: keep this many queries per image")
    ap.add_argument("--resume-dir", default=None)
    ap.add_argument("--warmup-images", type=int, default=2,
                    help="images explained BEFORE the clock starts (hipGraph capture, first-touch allocations); 0: time everything")
    ap.add_argument("--no-tuned-gemms", action="store_true", help="keep the default hipBLASLt / rocBLAS heuristic for the body's GEMMs")
    args = ap.parse_args()
    rank, world, dev, gather_dev = sharding.init_evaluator_process()
    if world > 1:
        import torch.distributed as dist
    from transformer_mm_explainability_amd import detr_model, tuned_gemms
    from transformer_mm_explainability_amd.detr_explainability import MaskGenerator

    if not args.no_tuned_gemms:
        tuned_gemms.enable("detr")     # pre-tuned library-GEMM selection for the 100-query decoder shapes (tuning stays off)
    torch.manual_seed(0)
    model = detr_model.detr_resnet50_head().to(dev).eval()
    mg = MaskGenerator(model, threshold=0.5, graph_slots=args.graph_slots or None)
    queries = [0]

    def masks_of(k):
        feats = synthetic_features(k, device=dev)
        with torch.no_grad():       # the 0.5 confidence cut of mask_generator.py:50 keeps nothing on random weights
            outputs = model(feats)
            conf = outputs["pred_logits"].softmax(-1)[0, :, :-1].max(-1).values
        mg.threshold = conf.sort().values[-args.keep_top - 1]                   # a device scalar: no host read for it
        # one forward per image (not two), ONE device -> host read per image (the keep mask, inside get_masks)
        masks, keep = mg.get_masks(feats, args.method, outputs=outputs)
        queries[0] += mg.last_kept
        return masks, keep

    cfg = {"evaluator": "detr_masks", "method": args.method, "keep_top": args.keep_top}
    store = sharding.PartialScores(args.resume_dir, rank, config=cfg) if args.resume_dir else None
    ids = list(range(args.num_images))          # DistributedSampler(shuffle=False): the dataset order
    for k in ids[:args.warmup_images]:          # one-time costs out of the rate: capture of the K-slot pass, slab allocation
        masks_of(k)
    queries[0] = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    table = evaluate(ids, masks_of, store=store, device=gather_dev)
    mg.check_diag()                              # the handle_residual word of the last image's passes
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=gather_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # The reference gathers variable-length per-rank pieces instead (pickled ``evalImgs`` + image ids: DETR/util/misc.py:88-128,
    # DETR/datasets/coco_eval.py:170-189).  Same pieces through that route -- this rank's image ids and its rows as an
    # ``[..., n_local_images]`` array -- must give the same table (outside the timed region; pycocotools' own content is out of reach here).
    mine = sharding.shard_indices(ids)
    merged_ids, merged = sharding.merge_eval_images(mine, table[mine].T.reshape(STAT_COLS, 1, len(mine)).cpu().numpy(), device=gather_dev)
    merge_ok = merged_ids.tolist() == ids and bool((torch.from_numpy(merged[:, 0, :].T.copy()) == table.cpu()).all())
    if rank == 0:
        print(json.dumps({"images": len(ids), "n_gpus": world, "eval_imgs_merge_matches": merge_ok, "seconds": round(elapsed, 3), "warmup_images": args.warmup_images,
                          "images_per_s": round(len(ids) / elapsed, 1),
                          "queries_per_s_this_rank": round(queries[0] / elapsed, 1), "method": args.method,
                          "mean_kept": round(float(table[:, 0].mean()), 2),
                          "mean_mask_area": round(float(table[:, 1].mean()), 4)}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
