"""Bisect helper (round 3): ViT-B/16 GraphedRelevance with the per-layer rule hooks under hipGraph capture.  python tools/bisect_vit_graph.py VARIANT"""
import sys

import torch

sys.path.insert(0, ".")
from transformer_mm_explainability_amd import ops, vit_model  # noqa: E402

variant = sys.argv[1]
if "norecord" in variant:
    torch.Tensor.record_stream = lambda self, stream: None
if "mainstream" in variant:
    ops.side_stream = lambda device, slot=0: torch.cuda.current_stream()
torch.manual_seed(0)
model = vit_model.vit_base_patch16_224().float().eval().cuda()
for p in model.parameters():
    p.requires_grad_(False)
x = torch.randn(1, 3, 224, 224, device="cuda")
K = 3 if "k3" in variant else 1
want = vit_model.generate_relevance_multi(model, x, list(range(K))).clone()
run = vit_model.GraphedRelevance(model, x, indices=list(range(K)))
print(variant, "captured", flush=True)
got = run(x)
torch.cuda.synchronize()
print(variant, "OK", float((got - want).abs().max()), flush=True)
