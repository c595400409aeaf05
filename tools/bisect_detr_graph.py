"""Bisect helper (round 3): which of the new DETR-pass features breaks a hipGraph replay after eager work.  Usage: python tools/bisect_detr_graph.py VARIANT"""
import sys

import torch

sys.path.insert(0, ".")
from transformer_mm_explainability_amd import attention_modules, detr_model, ops  # noqa: E402
from transformer_mm_explainability_amd.detr_explainability import Generator, GraphedGenerateOursMulti  # noqa: E402

variant = sys.argv[1]
if "norules" in variant:
    Generator.overlap_rules = False
if "novalue" in variant:
    attention_modules.MultiheadAttention.overlap_value_proj = False
if "norecord" in variant:
    torch.Tensor.record_stream = lambda self, stream: None
if "nosplit" in variant:
    ops.set_option("attn_fwd_split", 0)
torch.manual_seed(0)
model = detr_model.detr_resnet50_head().cuda().eval()
feats = torch.randn(1, 2048, 25, 38, device="cuda") * 0.5
t = torch.tensor([25, 33, 46, 49, 53, 60, 89, 95], device="cuda")
if "first" in variant:
    run = GraphedGenerateOursMulti(model, feats, K=8, rows_only=False)
    with torch.no_grad():
        model(feats)
    run(feats, t)
want = Generator(model).generate_ours_multi(feats, t)
run_rows = GraphedGenerateOursMulti(model, feats, K=8)
with torch.no_grad():
    model(feats)
out = run_rows(feats, t)
torch.cuda.synchronize()
print(variant, "OK", float((out - want).abs().max()))
