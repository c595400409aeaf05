"""Make ``tuning/tunableop_gfx950_<workload>.csv`` on an MI355X: run a workload's passes eagerly with PyTorch TunableOp
tuning ON (every distinct GEMM shape is timed against the hipBLASLt / rocBLAS solutions once), write the selection.

    python tools/tune_gemms.py detr|lxmert|lxmert_pert|vit_b16|cfg5 <out.csv>

``lxmert_pert``: the perturbation re-runs only (eager no-grad forwards at 8-9x the batch); rows below 2048 are dropped from the
file afterwards -- small shapes also occur in the CAPTURED explain pass, which must not meet a tuned solution.
"""
import os
import sys

work, out = sys.argv[1], sys.argv[2]
os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
os.environ["PYTORCH_TUNABLEOP_TUNING"] = "1"
os.environ["PYTORCH_TUNABLEOP_FILENAME"] = out
os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS", "15")
os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS", "5")
os.environ.setdefault("PYTORCH_TUNABLEOP_VERBOSE", "0")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.manual_seed(0)
if work == "detr":
    from transformer_mm_explainability_amd import detr_model
    from transformer_mm_explainability_amd.detr_explainability import Generator
    model = detr_model.detr_resnet50_head().cuda().eval()
    feats = torch.randn(1, 2048, 25, 38, device="cuda") * 0.5
    with torch.no_grad():
        model(feats)
    gen = Generator(model)
    for K in (8, 10, 16, 20):
        gen.generate_ours_multi(feats, torch.arange(K, device="cuda") * 3, rows_only=True)
    gen.generate_ours(feats, torch.tensor([3], device="cuda"), use_lrp=False)
elif work == "lxmert":
    from transformer_mm_explainability_amd import lxmert_explainability as le
    from transformer_mm_explainability_amd import lxmert_model as lm
    from transformer_mm_explainability_amd import lxmert_perturbation as lp
    cfg = lm.LxmertConfig()
    model = lm.LxmertForQuestionAnswering(cfg).cuda().eval()
    gen = le.GeneratorOurs(type("Usage", (), {"model": model})())
    pert = lp.LxmertPerturbation(model)
    for B in (8, 32):
        T = 20
        batch = dict(input_ids=torch.randint(1, cfg.vocab_size, (B, T), device="cuda"), attention_mask=torch.ones(B, T, device="cuda"),
                     token_type_ids=torch.zeros(B, T, dtype=torch.long, device="cuda"),
                     visual_feats=torch.randn(B, 36, cfg.visual_feat_dim, device="cuda"), visual_pos=torch.rand(B, 36, 4, device="cuda"))
        R_t_t, R_t_i = gen.generate_ours_batch(batch)
        cam_image, cam_text = lp.normalize_cams_batch(R_t_t, R_t_i, batch["attention_mask"])
        pert.perturbation_image(batch, cam_image, False)
        pert.perturbation_text(batch, cam_text, False)
elif work == "lxmert_pert":
    from transformer_mm_explainability_amd import lxmert_model as lm
    from transformer_mm_explainability_amd import lxmert_perturbation as lp
    cfg = lm.LxmertConfig()
    model = lm.LxmertForQuestionAnswering(cfg).cuda().eval()
    pert = lp.LxmertPerturbation(model, tuned=False)
    B = 32
    for T in (14, 20):            # bench.py's cfg-4 leg (T = 14) and the evaluator's padded questions (T = 20)
        batch = dict(input_ids=torch.randint(1, cfg.vocab_size, (B, T), device="cuda"), attention_mask=torch.ones(B, T, device="cuda"),
                     token_type_ids=torch.zeros(B, T, dtype=torch.long, device="cuda"),
                     visual_feats=torch.randn(B, 36, cfg.visual_feat_dim, device="cuda"), visual_pos=torch.rand(B, 36, 4, device="cuda"))
        pert.perturbation_image(batch, torch.rand(B, 36, device="cuda"), False)
        if T == 20:
            pert.perturbation_text(batch, torch.rand(B, T, device="cuda"), False)
elif work == "vit_b16":
    from transformer_mm_explainability_amd import vit_model
    model = vit_model.vit_base_patch16_224().float().eval().cuda()
    for p in model.parameters():
        p.requires_grad_(False)
    x = torch.randn(1, 3, 224, 224, device="cuda")
    for K in (1, 8):              # one image: 197-row forward; 197- / 1576-row backward (bench.py's cfg-1 leg: 1 and 8 targets)
        vit_model.generate_relevance_multi(model, x, indices=list(range(K)))
elif work == "cfg5":
    from transformer_mm_explainability_amd import clip_explainability as ce
    from transformer_mm_explainability_amd import clip_model
    import bench
    model = clip_model.random_init("ViT-L/14@336", seed=0).cuda()
    model.set_body_dtype(torch.bfloat16)
    image = torch.randn(1, 3, 336, 336, device="cuda")
    _, texts = bench.synthetic_inputs(128, "cuda", 0)
    ce.interpret(image, texts, model, "cuda", 0, 0)
else:
    raise SystemExit("unknown workload " + work)
torch.cuda.synchronize()
# this PyTorch streams the results into <stem><device ordinal><ext> as they are found (there is no write_file any more)
stem, ext = os.path.splitext(out)
rows = open(stem + "0" + ext).read().splitlines()
if work == "lxmert_pert":
    keep = [r for r in rows if r.startswith("Validator") or int(r.split(",")[1].split("_")[2]) >= 2048]
    print("kept %d of %d entries (rows >= 2048)" % (len(keep), len(rows)))
    rows = keep
open(out, "w").write("\n".join(rows) + "\n")
print("wrote", out)
