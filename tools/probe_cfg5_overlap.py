import sys, time, torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import bench_legs as bl
from transformer_mm_explainability_amd import clip_explainability as ce
model, image, texts, attn_layer, _ = bl.cfg5_setup(128, torch.device("cuda"))
def t(fn, reps=4, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for ov in (True, False, True, False):
    print("overlap_towers=%s: %.2f ms" % (ov, t(lambda: ce.interpret(image, texts, model, "cuda", start_layer=0, start_layer_text=0, overlap_towers=ov))), flush=True)
