"""cfg 1 alone for a rocprofv3 kernel trace: ViT-B/16, one image, one target, replayed from the hipGraph.
    rocprofv3 --kernel-trace --stats -d out -o vit -- python tools/probe_vit_trace.py 20"""
import sys

import torch

sys.path.insert(0, ".")
from transformer_mm_explainability_amd import vit_model  # noqa: E402

reps = int(sys.argv[1])
torch.manual_seed(0)
model = vit_model.vit_base_patch16_224().float().eval().cuda()
for p in model.parameters():
    p.requires_grad_(False)
x = torch.randn(1, 3, 224, 224, device="cuda")
run = vit_model.GraphedRelevance(model, x, indices=[5])
run(x)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    run(x)
b.record()
torch.cuda.synchronize()
print("ViT-B/16 one image, one target, hipGraph replay: %.3f ms" % (a.elapsed_time(b) / reps), file=sys.stderr)
