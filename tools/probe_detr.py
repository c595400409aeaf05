"""DETR-R50 head (config 3): per-query latency of Generator.generate_ours at 25x38 image tokens, 100 queries."""
import sys
import time

import torch

sys.path.insert(0, ".")
from transformer_mm_explainability_amd import detr_model  # noqa: E402
from transformer_mm_explainability_amd.detr_explainability import Generator  # noqa: E402

import os as _os
if _os.environ.get("MMX_TUNED", "1") == "1":
    from transformer_mm_explainability_amd import tuned_gemms
    print("tuned GEMM selection loaded:", tuned_gemms.enable("detr"))
torch.manual_seed(0)
model = detr_model.detr_resnet50_head().cuda().eval()
feats = torch.randn(1, 2048, 25, 38, device="cuda") * 0.5
gen = Generator(model)


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


tgt = torch.tensor([5], device="cuda")
print("forward only        %.2f ms" % timed(lambda: model(feats)))
print("generate_ours/query %.2f ms" % timed(lambda: gen.generate_ours(feats, tgt, use_lrp=False)))
print("generate_rollout    %.2f ms" % timed(lambda: gen.generate_rollout(feats, tgt)))


def fwd_bwd():
    out = model(feats)["pred_logits"]
    model.zero_grad()
    out[0, 5, 3].backward()


print("forward+backward    %.2f ms" % timed(fwd_bwd))

for K in (1, 5, 10, 20):
    t = torch.arange(K, device="cuda") * 3
    ms = timed(lambda: gen.generate_ours_multi(feats, t), n=5, warm=2)
    print("generate_ours_multi K=%-2d %.2f ms  (%.2f ms/query, %.0f queries/s)" % (K, ms, ms / K, K / ms * 1e3))

from transformer_mm_explainability_amd.detr_explainability import GraphedGenerateOursMulti, MaskGenerator  # noqa: E402
for rows_only in (False, True):
    for K in (10, 16, 20):
        run = GraphedGenerateOursMulti(model, feats, K=K, rows_only=rows_only)
        t = torch.arange(K, device="cuda") * 3
        ms = timed(lambda: run(feats, t), n=10, warm=2)
        print("generate_ours_multi hipGraph K=%-2d %s %.2f ms  (%.2f ms/query, %.0f queries/s; one deferred diag read per call)"
              % (K, "row-vector rules" if rows_only else "matrix rules    ", ms, ms / K, K / ms * 1e3))
        del run
for K in (10, 20):
    t = torch.arange(K, device="cuda") * 3
    ms = timed(lambda: gen.generate_ours_multi(feats, t, rows_only=True), n=5, warm=2)
    print("generate_ours_multi eager, row-vector rules K=%-2d %.2f ms  (%.2f ms/query)" % (K, ms, ms / K))

for K in (5, 10, 20):
    t = torch.arange(K, device="cuda") * 3
    ms = timed(lambda: gen.generate_ours_multi(feats, t, share_forward=False), n=5, warm=2)
    print("  replicated forward       K=%-2d %.2f ms  (%.2f ms/query)" % (K, ms, ms / K))

model.transformer.backward_gemm_dtype = torch.bfloat16
for K in (10, 20):
    t = torch.arange(K, device="cuda") * 3
    ms = timed(lambda: gen.generate_ours_multi(feats, t), n=5, warm=2)
    print("  bf16 backward GEMMs      K=%-2d %.2f ms  (%.2f ms/query, %.0f queries/s)" % (K, ms, ms / K, K / ms * 1e3))
