"""GPU-box probe: what the LRP route (``use_lrp=True``, the generators' DEFAULT argument) costs next to the
no-LRP route on the measured bodies -- DETR-R50 head at 950 image tokens (one kept query per call, as
``DETR/mask_generator.py:90-110`` runs it) and LXMERT-base at T = 14 / I = 36 (one item per call, as ``perturbation.py:216-238``).
``python tools/probe_lrp.py [detr|lxmert|both] [reps]``; run under ``rocprofv3 --kernel-trace --stats`` for the kernel table."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import types
import torch


def timed(fn, reps, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def detr(reps):
    from transformer_mm_explainability_amd import detr_model
    from transformer_mm_explainability_amd.detr_explainability import Generator
    torch.manual_seed(0)
    model = detr_model.detr_resnet50_head().cuda().eval()
    feats = torch.randn(1, 2048, 25, 38, device="cuda") * 0.5
    gen = Generator(model)
    t = torch.tensor([3], device="cuda")
    no = timed(lambda: gen.generate_ours(feats, t, use_lrp=False), reps)
    yes = timed(lambda: gen.generate_ours(feats, t), reps)
    att = timed(lambda: gen.generate_transformer_att(feats, t), reps)
    print("DETR-R50 head, 950 tokens, one query per call: ours_no_lrp %.2f ms | ours (use_lrp=True) %.2f ms (%.2fx) | "
          "transformer_att %.2f ms" % (no, yes, yes / no, att))
    return {"ours_no_lrp_ms": round(no, 3), "ours_lrp_ms": round(yes, 3), "ratio": round(yes / no, 3),
            "transformer_att_ms": round(att, 3)}


def lxmert(reps):
    from transformer_mm_explainability_amd import lxmert_explainability as le
    from transformer_mm_explainability_amd import lxmert_model as lm
    torch.manual_seed(0)
    model = lm.LxmertForQuestionAnswering(lm.LxmertConfig()).cuda().eval()
    T, I = 14, 36
    gb = torch.Generator().manual_seed(2)
    item = dict(input_ids=torch.randint(1, 30000, (1, T), generator=gb).cuda(), attention_mask=torch.ones(1, T).cuda(),
                token_type_ids=torch.zeros(1, T, dtype=torch.long).cuda(),
                visual_feats=torch.randn(1, I, 2048, generator=gb).cuda(), visual_pos=torch.rand(1, I, 4, generator=gb).cuda())
    usage = types.SimpleNamespace(model=model, text_len=T, image_boxes_len=I, forward=lambda it: model(**item))
    gen = le.GeneratorOurs(usage)
    no = timed(lambda: gen.generate_ours(None, use_lrp=False), reps)
    yes = timed(lambda: gen.generate_ours(None), reps)
    print("LXMERT-base, T = 14, I = 36, one item per call: ours_no_lrp %.2f ms | ours (use_lrp=True) %.2f ms (%.2fx)"
          % (no, yes, yes / no))
    return {"ours_no_lrp_ms": round(no, 3), "ours_lrp_ms": round(yes, 3), "ratio": round(yes / no, 3)}


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    out = {}
    if which in ("detr", "both"):
        out["detr"] = detr(reps)
    if which in ("lxmert", "both"):
        out["lxmert"] = lxmert(reps)
    import json
    print("LRP_PROBE_JSON " + json.dumps(out))
