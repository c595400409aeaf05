"""Kernel-trace target: CLIP ViT-L/14@336 interpret, batch 32, the bf16 body of clip_model.CLIP.set_body_dtype (3 steps)."""
import sys

import torch

sys.path.insert(0, ".")
from transformer_mm_explainability_amd import clip_explainability as ce  # noqa: E402
from transformer_mm_explainability_amd import clip_model  # noqa: E402

dev = torch.device("cuda")
model = clip_model.random_init("ViT-L/14@336", seed=0).to(dev)
model.set_body_dtype(torch.bfloat16)
image = torch.randn(1, 3, 336, 336, device=dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
texts = torch.zeros(B, 77, dtype=torch.long)
texts[:, 0] = 49406
texts[:, 1:6] = 1000
texts[:, 6] = 49407
texts = texts.to(dev)
for _ in range(3):
    ce.interpret(image, texts, model, dev, start_layer=0, start_layer_text=0)
torch.cuda.synchronize()
