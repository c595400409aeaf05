"""cfg 4's perturbation leg alone (B = 32, image test: the 9 re-runs as ONE no-grad tape forward) for a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --stats -d out -o pert -- python tools/probe_lxmert_pert_trace.py 32 5"""
import sys

import torch

sys.path.insert(0, ".")
from transformer_mm_explainability_amd import lxmert_model as lm  # noqa: E402
from transformer_mm_explainability_amd import lxmert_perturbation as lp  # noqa: E402

B, reps = int(sys.argv[1]), int(sys.argv[2])
torch.manual_seed(0)
model = lm.LxmertForQuestionAnswering(lm.LxmertConfig()).cuda().eval()
T, I = 14, 36
g = torch.Generator().manual_seed(2)
batch = dict(input_ids=torch.randint(1, 30000, (B, T), generator=g).cuda(), attention_mask=torch.ones(B, T).cuda(),
             token_type_ids=torch.zeros(B, T, dtype=torch.long).cuda(),
             visual_feats=torch.randn(B, I, 2048, generator=g).cuda(), visual_pos=torch.rand(B, I, 4, generator=g).cuda())
cams = torch.rand(B, I, generator=g).cuda()
pert = lp.LxmertPerturbation(model, tuned=(len(sys.argv) < 4 or sys.argv[3] != "untuned"))
for _ in range(2):
    pert.perturbation_image(batch, cams)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    pert.perturbation_image(batch, cams)
b.record()
torch.cuda.synchronize()
print("B=%d image test (%s): %.2f ms per call" % (B, "tuned selection if shipped" if pert.tuned else "default heuristic", a.elapsed_time(b) / reps), file=sys.stderr)
