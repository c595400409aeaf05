"""BASELINE.json config 5 shape (CLIP ViT-L/14@336: image tower 24 x 16 heads x 577 tokens, text 12 x 12 x 77) on the
fp32 path: self-consistency of the shared-forward route at N = 577, then maps/s at growing batch."""
import sys
import time

import torch

sys.path.insert(0, ".")
from transformer_mm_explainability_amd import clip_explainability as ce  # noqa: E402
from transformer_mm_explainability_amd import clip_model  # noqa: E402
import os as _os  # noqa: E402
if _os.environ.get("MMX_TUNED", "1") == "1":
    from transformer_mm_explainability_amd import tuned_gemms  # noqa: E402
    print("tuned GEMM selection loaded:", tuned_gemms.enable("clip_vitl14_336_bf16"))

dev = torch.device("cuda")
model = clip_model.random_init("ViT-L/14@336", seed=0).to(dev)
g = torch.Generator().manual_seed(1)
image = torch.randn(1, 3, 336, 336, generator=g).to(dev)


def prompts(B):
    t = torch.zeros(B, 77, dtype=torch.long)
    gg = torch.Generator().manual_seed(2)
    for b in range(B):
        n = int(torch.randint(3, 11, (1,), generator=gg))
        t[b, 0] = 49406
        t[b, 1:1 + n] = torch.randint(1, 49405, (n,), generator=gg)
        t[b, 1 + n] = 49407
    return t.to(dev)


texts = prompts(2)
a = ce.interpret(image, texts, model, dev, start_layer=0, start_layer_text=0, share_image_forward=True)
b = ce.interpret(image, texts, model, dev, start_layer=0, start_layer_text=0, share_image_forward=False)
print("shared vs replicated image tower, B=2: max |dR_image| = %.3e (max |R| = %.3e), max |dR_text| = %.3e"
      % ((a[1] - b[1]).abs().max().item(), b[1].abs().max().item(), (a[0] - b[0]).abs().max().item()))
import sys as _sys  # noqa: E402
if len(_sys.argv) > 1 and _sys.argv[1] == "bf16":
    model.visual.transformer.capture_dtype = torch.bfloat16      # image tower slabs in bf16 (config 5's "bf16 capture")
    print("image-tower capture slabs: bf16")
if len(_sys.argv) > 2 and _sys.argv[2] == "gemm":
    ref = ce.interpret(image, prompts(4), model, dev, start_layer=0, start_layer_text=0)[1].clone()
    model.visual.transformer.backward_gemm_dtype = torch.bfloat16
    got = ce.interpret(image, prompts(4), model, dev, start_layer=0, start_layer_text=0)[1]
    cos = torch.nn.functional.cosine_similarity(got, ref, dim=-1)
    print("backward GEMMs in bf16: min cosine similarity of the image maps vs fp32 GEMMs = %.6f, max rel err = %.3e"
          % (cos.min().item(), ((got - ref).abs().max() / ref.abs().max()).item()))
if len(_sys.argv) > 3 and _sys.argv[3] == "body":
    ref = ce.interpret(image, prompts(4), model, dev, start_layer=0, start_layer_text=0)
    ref = [r.clone() for r in ref]
    model.set_body_dtype(torch.bfloat16)
    got = ce.interpret(image, prompts(4), model, dev, start_layer=0, start_layer_text=0)
    for name, a_, b_ in (("text", got[0], ref[0]), ("image", got[1], ref[1])):
        cos = torch.nn.functional.cosine_similarity(a_, b_, dim=-1)
        print("bf16 body (all GEMMs + image attention on the bf16 matrix cores) vs the previous setting, %s maps: min cosine "
              "= %.6f, max rel err = %.3e" % (name, cos.min().item(), ((a_ - b_).abs().max() / b_.abs().max()).item()))
for B in ((128,) if len(_sys.argv) > 1 else (16, 64, 128)):
    texts = prompts(B)
    torch.cuda.reset_peak_memory_stats()
    for _ in range(2):
        ce.interpret(image, texts, model, dev, start_layer=0, start_layer_text=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        ce.interpret(image, texts, model, dev, start_layer=0, start_layer_text=0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("ViT-L/14@336 fp32, B=%-3d all layers: %.1f ms/step = %.1f maps/s, peak memory %.1f GB"
          % (B, dt * 1e3, B / dt, torch.cuda.max_memory_allocated() / 2**30))
