"""Compose ``profiles/<tag>_cfg_legs.txt`` from one GPU round: the ``configs`` object of the bench line
(``gpurun_out/<tag>/bench.json``), a reading of each leg generated from those numbers, and the kernel table of the same default
command under rocprofv3 (``gpurun_out/<tag>/cfg_legs.txt``, made by ``tools/gpu_round.sh``).

    python tools/make_cfg_legs.py r03
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag)
line = [ln for ln in open(os.path.join(src, "bench.json")).read().splitlines() if ln.startswith("{")][-1]
cfg = json.loads(line)["configs"]
out = ["# The four BASELINE configurations next to the headline (cfg 2), as `configs.*` of the ONE bench line `python bench.py` prints",
       "# (tools/bench_legs.py; tools/gpu_round.sh %s -> profiles/%s_bench.json; this file: tools/make_cfg_legs.py %s).  Each leg: rate, ms, the" % (tag, tag, tag),
       "# dominant kernel of OUR part with its algorithmic bytes / flop per launch, HIP-event us per launch and roofline fraction.", "#"]
for name in sorted(cfg):
    out += ["## " + name, json.dumps(cfg[name], indent=1)]
c1, c3, c4, c5 = (cfg.get(k, {}) for k in ("cfg1", "cfg3", "cfg4", "cfg5"))
out += ["#", "# Reading the legs:"]
if c1.get("ms"):
    mt = c1.get("multi_target", {})
    out.append("#  cfg1  ViT-B/16, ONE image, one target: %.2f ms per map from a hipGraph = %.0f maps/s; %s targets of one image share the forward: %s maps/s."
               % (c1["ms"], c1["rate"], mt.get("targets", "?"), mt.get("rate", mt.get("maps_per_s", "?"))))
if c3.get("ms"):
    ev = c3.get("evaluator", {})
    out.append("#  cfg3  DETR-R50 head, 950 image tokens: K = 10 kept queries %.2f ms, K = 20 %.2f ms = %.0f queries/s; the --masks evaluator loop (forward, keep set,"
               % (c3["K10"]["ms"], c3["K20"]["ms"], c3["rate"]))
    out.append("#        one %s-slot pass, Otsu masks, ONE device->host read per image) %.2f ms per image = %.0f images/s; steps in profiles/%s_detr_probe.txt."
               % (ev.get("kept_queries_per_image", "?"), ev.get("ms_per_image", float("nan")), ev.get("images_per_s", float("nan")), tag))
if c4.get("ms"):
    out.append("#  cfg4  LXMERT B = 32: explain %.2f ms (two modalities side by side, one hipGraph) + the 9-step image perturbation %.1f ms (keep masks for the"
               % (c4["explain_ms"], c4["perturb_ms"]))
    out.append("#        whole batch at once, the text's own layers once per sample, tuned GEMM selection scoped to the re-runs) = %.0f samples/s; profiles/%s_lxmert_probe.txt."
               % (c4["rate"], tag))
if c5.get("ms"):
    out.append("#  cfg5  CLIP ViT-L/14@336 bf16 body, batch 128: %.1f ms = %.0f maps/s (text tower on the captions' own length, exact: %s ms); profiles/%s_cfg5_probe.txt."
               % (c5["ms"], c5["rate"], c5.get("variant_trim_text_padding", {}).get("ms", "?"), tag))
table = os.path.join(src, "cfg_legs.txt")
if os.path.exists(table):
    out += ["#", "# ---- kernels of the DEFAULT command's legs (rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline;",
            "#      headline + variants + the four legs in one process; tools/prof_summary.py --by-grid, our kernels only):"]
    out += open(table).read().splitlines()
dst = os.path.join(ROOT, "profiles", "%s_cfg_legs.txt" % tag)
open(dst, "w").write("\n".join(out) + "\n")
print("wrote", dst)
