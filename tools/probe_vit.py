"""Probe (GPU box): ViT-B/16 (BASELINE config 1 architecture, random init) relevancy maps: GPU path vs CPU oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer_mm_explainability_amd import vit_model, vit_explainability as ve
from oracle import vit_torch

torch.manual_seed(0)
model = vit_model.vit_base_patch16_224().float().eval()
x = torch.randn(1, 3, 224, 224)
sd = {k: v.clone() for k, v in model.state_dict().items()}
torch.set_num_threads(min(os.cpu_count() or 1, 32))
vit_torch.generate_relevance(sd, x, 12, 5)
t0 = time.perf_counter(); want, _ = vit_torch.generate_relevance(sd, x, 12, 5); cpu = time.perf_counter() - t0
print(f"ViT-B/16 CPU oracle (reference algorithm, {torch.get_num_threads()} threads): {cpu*1e3:.1f} ms per map = {1/cpu:.2f} maps/s")
model = model.cuda(); xc = x.cuda()
for p in model.parameters(): p.requires_grad_(False)
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
got = ve.generate_relevance(model, xc.requires_grad_(True), index=5)
print("max |gpu - oracle| =", (got.cpu() - want).abs().max().item())
t1 = timed(lambda: ve.generate_relevance(model, xc, index=5))
print(f"ViT-B/16 MI355X generate_relevance (1 target): {t1*1e3:.2f} ms per map = {1/t1:.1f} maps/s")
for K in (2, 8, 32):
    tk = timed(lambda: vit_model.generate_relevance_multi(model, xc, list(range(K))))
    print(f"ViT-B/16 MI355X generate_relevance_multi K={K} targets, one forward: {tk*1e3:.2f} ms = {K/tk:.1f} maps/s")
for K in (1, 8):
    run = vit_model.GraphedRelevance(model, xc, indices=list(range(K)))
    tk = timed(lambda: run(xc), n=50)
    print(f"ViT-B/16 MI355X GraphedRelevance (hipGraph replay) K={K}: {tk*1e3:.2f} ms = {K/tk:.1f} maps/s")
model.backward_gemm_dtype = torch.bfloat16
for K in (32,):
    tk = timed(lambda: vit_model.generate_relevance_multi(model, xc, list(range(K))))
    print(f"ViT-B/16 MI355X generate_relevance_multi K={K}, bf16 backward GEMMs: {tk*1e3:.2f} ms = {K/tk:.1f} maps/s")
