"""Probe (GPU box, under rocprofv3 --kernel-trace --stats): a few cfg-5 steps (CLIP ViT-L/14@336, bf16 body, batch 128) for the
per-kernel split of the step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import bench_legs  # noqa: E402
from transformer_mm_explainability_amd import clip_explainability as ce  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
model, image, texts, _, _ = bench_legs.cfg5_setup(B, torch.device("cuda"))
for _ in range(steps):
    ce.interpret(image, texts, model, "cuda", start_layer=0, start_layer_text=0)
torch.cuda.synchronize()
