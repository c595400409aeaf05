"""Run N steady-state interpret() steps only (for a clean per-step kernel trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from transformer_mm_explainability_amd import clip_explainability as ce, clip_model
model = clip_model.random_init("ViT-B/32", 0).cuda()
image, texts = bench.synthetic_inputs(64, "cuda", 0)
for _ in range(13):
    ce.interpret(image, texts, model, "cuda", 0, 0)
torch.cuda.synchronize()
