"""BUILD-CONTAINER ONLY (reads /root/reference): how fast is the CPU *port* (oracle/clip_torch.py, ``cpu_baseline.kind = "port"``) next to
the REFERENCE ITSELF -- CLIP/clip/model.py imported by file path + ``interpret`` exec'd from CLIP_explainability.ipynb cell 6, exactly
as tests/golden/make_golden.py loads them -- on the same weights (clip_model.random_init("ViT-B/32", seed 0)), the same synthetic
inputs (bench.py's), the same thread count?  VERDICT r04 missing #6.  Writes profiles/r05_cpu_port_calibration.txt.

    python tools/cpu_port_calibration.py [batch] [reps]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import torch


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    import make_golden as mg
    from oracle import clip_torch
    from transformer_mm_explainability_amd import clip_model
    import bench
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    model_mod, _, _, _, _ = mg._clip_tiny_reference()
    sd = clip_model.random_init("ViT-B/32", seed=0).state_dict()
    cfg = dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768, vision_patch_size=32,
               context_length=77, vocab_size=49408, transformer_width=512, transformer_heads=8, transformer_layers=12)
    ref = model_mod.CLIP(**cfg).float().eval()
    ref.load_state_dict(sd)
    ns = {"torch": torch, "np": np, "start_layer": -1, "start_layer_text": -1}
    exec(mg.notebook_cell("CLIP_explainability.ipynb", 6), ns)
    image, texts = bench.synthetic_inputs(batch, "cpu", 0)
    port_sd = clip_torch.prepare_state_dict(sd, 8)

    def run_ref():
        return ns["interpret"](image, texts, ref, "cpu", start_layer=0, start_layer_text=0)

    def run_port(split=None):
        return clip_torch.interpret(port_sd, image, texts, 0, 0, timings=split)

    def median(fn):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            out = fn()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2], out

    t_ref, (rt_ref, ri_ref) = median(run_ref)
    split = {}
    t_port, (rt_port, ri_port) = median(lambda: run_port(split))
    lines = [
        "# CPU port calibration (VERDICT r04 missing #6), build container: %d host cores, torch %s, %d threads" % (cores, torch.__version__, cores),
        "# workload: CLIP ViT-B/32 random-init (seed 0), bench.py's synthetic image + %d captions, all 12 + 12 layers (start_layer = 0), fp32" % batch,
        "# reference = /root/reference/CLIP/clip/model.py (file-path import) + CLIP_explainability.ipynb cell 6 `interpret` (exec'd from the notebook JSON)",
        "# port      = oracle/clip_torch.interpret (what bench.py's cpu_baseline times on the GPU box, where /root/reference does not exist)",
        "reference : %.3f s per call = %.2f maps/s   (median of %d)" % (t_ref, batch / t_ref, reps),
        "port      : %.3f s per call = %.2f maps/s   (median of %d)   port / reference time = %.3f" % (t_port, batch / t_port, reps, t_port / t_ref),
        "port split of its last run: forward %.3f s | 24 per-layer partial backwards %.3f s | rule chain %.3f s" %
        (split["forward_s"], split["backward_s"], split["rules_s"]),
        "results   : max |R_text port - reference| = %.3e (max |ref| %.3e), max |R_image port - reference| = %.3e (max |ref| %.3e)" %
        (float((rt_port - rt_ref).abs().max()), float(rt_ref.abs().max()), float((ri_port - ri_ref).abs().max()), float(ri_ref.abs().max())),
    ]
    text = "\n".join(lines) + "\n"
    print(text)
    with open(os.path.join(ROOT, "profiles", "r05_cpu_port_calibration.txt"), "w") as f:
        f.write(text)


if __name__ == "__main__":
    main()
