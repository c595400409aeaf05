import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # enables tuned GEMMs unless PYTORCH_TUNABLEOP_ENABLED is preset
import torch
from transformer_mm_explainability_amd import clip_explainability as ce, clip_model
model = clip_model.random_init("ViT-B/32", 0).cuda()
image, texts = bench.synthetic_inputs(64, "cuda", 0)
print("max EOT+1 =", int(texts.argmax(-1).max()) + 1)
for trim in (False, True):
    ts = []
    for i in range(12):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ce.interpret(image, texts, model, "cuda", 0, 0, trim_text_padding=trim)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("trim", trim, " ".join("%.1f" % t for t in ts))
