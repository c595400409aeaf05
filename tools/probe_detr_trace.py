"""Kernel trace target: 10 x Generator.generate_ours on the DETR-R50 head (run under rocprofv3 --kernel-trace)."""
import sys

import torch

sys.path.insert(0, ".")
from transformer_mm_explainability_amd import detr_model  # noqa: E402
from transformer_mm_explainability_amd.detr_explainability import Generator  # noqa: E402

torch.manual_seed(0)
model = detr_model.detr_resnet50_head().cuda().eval()
feats = torch.randn(1, 2048, 25, 38, device="cuda") * 0.5
gen = Generator(model)
tgt = torch.tensor([5], device="cuda")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    gen.generate_ours(feats, tgt, use_lrp=False)
torch.cuda.synchronize()
