"""CLIP ViT-B/32 step (B = 64, all layers, hipGraph replay) with the whole-head attention kernels vs the streaming ones."""
import sys
import time

sys.path.insert(0, ".")
import bench  # noqa: E402  (enables the tuned GEMM selection before torch is imported)
import torch  # noqa: E402

from transformer_mm_explainability_amd import clip_explainability as ce  # noqa: E402
from transformer_mm_explainability_amd import clip_model, ops  # noqa: E402

dev = torch.device("cuda")
model = clip_model.random_init("ViT-B/32", seed=0).to(dev)
image, texts = bench.synthetic_inputs(64, dev, seed=0)
for small in (1, 0):
    ops.set_option("attn_small", small)
    run = ce.GraphedInterpret(model, image, texts, start_layer=0, start_layer_text=0)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print("attn_small=%d: %.3f ms/step = %.1f maps/s" % (small, ms, 64 / ms * 1e3))
    del run
ops.set_option("attn_small", 1)
