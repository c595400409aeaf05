"""The four BASELINE.json configurations besides the headline (cfg 2), as bounded legs of ``bench.py``.

Each leg builds the configuration's architecture with random-init weights and synthetic inputs (there are no checkpoints
or datasets offline), times the end-to-end relevancy pass the way ``bench.py`` times the headline (warm-up, then a
synchronised loop), and times the dominant hand-written kernel of that pass STAND-ALONE with HIP events on the launch
stream.  A leg returns one JSON-serialisable dict:

    {"workload": ..., "rate": R, "unit": ..., "ms": T,                     # end to end
     "kernel": {"name", "bound": "hbm"|"mfma", "bytes_per_launch" | "flop_per_launch", "us_per_launch", "achieved", "peak",
                "unit", "frac"},                                            # the dominant kernel of OUR part of the pass
     "source": "profiles/..."}                                              # where the matching rocprof summary lives

``run_all`` never raises: a failing leg is reported as ``{"error": ...}`` so the headline line is still printed.
SURVEY.md section 8(d) config table is the spec for sizes."""
from __future__ import annotations

import time
import traceback

import torch

HBM_PEAK_GBS = 8000.0
FP32_MFMA_PEAK_TFLOPS = 157.3
BF16_MFMA_PEAK_TFLOPS = 2500.0


def _timed_ms(fn, n, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def _hbm(name, nbytes, us, note=None):
    ach = nbytes / us / 1e3
    d = {"name": name, "bound": "hbm", "bytes_per_launch": int(nbytes), "us_per_launch": round(us, 2),
         "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4)}
    if note:
        d["note"] = note
    return d


def _mfma(name, flop, us, peak, note=None):
    ach = flop / us / 1e6
    d = {"name": name, "bound": "mfma", "flop_per_launch": int(flop), "us_per_launch": round(us, 2),
         "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4)}
    if note:
        d["note"] = note
    return d


def leg_cfg1(kernel_time_us, reps=20):
    """cfg 1: ViT-B/16 (12 x 12 heads x 197 tokens), one image, class-index relevancy (ViT notebook cell 7:14-34)."""
    from transformer_mm_explainability_amd import ops, vit_model
    torch.manual_seed(0)
    model = vit_model.vit_base_patch16_224().float().eval().cuda()
    for p in model.parameters():
        p.requires_grad_(False)
    x = torch.randn(1, 3, 224, 224, device="cuda")
    run1 = vit_model.GraphedRelevance(model, x, indices=[5])
    ms1 = _timed_ms(lambda: run1(x), reps)
    run8 = vit_model.GraphedRelevance(model, x, indices=list(range(8)))
    ms8 = _timed_ms(lambda: run8(x), reps)
    # dominant rule kernel: the head average of one layer's slabs (the row-vector chain reads A-bar once per layer)
    H, N = 12, 197
    a = torch.softmax(torch.randn(1, H, N, N, device="cuda"), -1)
    g = torch.randn(1, H, N, N, device="cuda") * 1e-2
    us = kernel_time_us(lambda: ops.avg_heads(a, g, 1), 50, torch.cuda.current_stream())
    del run1, run8
    return {"workload": "BASELINE config 1 architecture: ViT-B/16, one 224x224 image, class-index relevancy over all 12 layers "
                        "(generate_relevance), fp32, replayed from a hipGraph",
            "rate": round(1e3 / ms1, 1), "unit": "maps/s", "ms": round(ms1, 3),
            "multi_target": {"targets": 8, "ms": round(ms8, 3), "rate": round(8e3 / ms8, 1)},
            "kernel": _hbm("avg_heads_kernel<f32> (one layer, B = 1: 12 heads x 197^2)", 2 * H * N * N * 4 + N * N * 4, us,
                           "latency-bound at batch 1 (3.7 MB per launch); the pass itself is ~700 body launches"),
            "source": "measured in this run (HIP events); rocprofv3 kernel table of the same legs: profiles/r06_cfg_legs.txt"}


def leg_cfg3(kernel_time_us, reps=10):
    """cfg 3: DETR-R50 transformer + heads at a 25 x 38 feature map (950 image tokens, 100 queries), K kept queries per
    image explained in one pass (DETR/mask_generator.py:90-121 loops over them)."""
    from transformer_mm_explainability_amd import detr_model, ops
    from transformer_mm_explainability_amd.detr_explainability import GraphedGenerateOursMulti
    torch.manual_seed(0)
    model = detr_model.detr_resnet50_head().cuda().eval()
    feats = torch.randn(1, 2048, 25, 38, device="cuda") * 0.5
    out = {}
    for K in (10, 20):
        run = GraphedGenerateOursMulti(model, feats, K=K)
        t = torch.arange(K, device="cuda") * 3
        ms = _timed_ms(lambda: run(feats, t), reps)
        out[K] = ms
        del run
    # the --masks evaluator's inner loop (DETR/engine.py:153-215 around MaskGenerator.get_masks): per image one forward, the keep
    # set, ONE K-slot pass for all kept queries, Otsu masks -- one device -> host read per image (examples/detr_masks_eval.py)
    from transformer_mm_explainability_amd.detr_explainability import MaskGenerator
    mg = MaskGenerator(model, threshold=0.5, graph_slots=8)
    keep_top, n_img = 8, 40

    def one_image(k):
        f = torch.randn(1, 2048, 25, 38, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5000 + k)) * 0.5
        with torch.no_grad():
            outputs = model(f)
            conf = outputs["pred_logits"].softmax(-1)[0, :, :-1].max(-1).values
        mg.threshold = conf.sort().values[-keep_top - 1]            # random weights: flat logits, keep the 8 most confident
        return mg.get_masks(f, "ours_no_lrp", outputs=outputs)[0]

    for k in range(2):
        one_image(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(2, 2 + n_img):
        one_image(k)
    mg.check_diag()
    torch.cuda.synchronize()
    ev_ms = (time.perf_counter() - t0) / n_img * 1e3
    del mg
    # the LRP route (use_lrp=True, the generators' DEFAULT argument): one kept query per call, as mask_generator.py:90-110 runs it
    from transformer_mm_explainability_amd.detr_explainability import Generator
    gen, t1 = Generator(model), torch.tensor([3], device="cuda")
    lrp_no = _timed_ms(lambda: gen.generate_ours(feats, t1, use_lrp=False), 3)
    lrp_yes = _timed_ms(lambda: gen.generate_ours(feats, t1), 3)
    del gen
    # dominant kernels of OUR part: the encoder self-attention backward pair (d = 32 streaming kernels) at K = 10
    K, H, N, D = 10, 8, 950, 32
    q, k, v = (torch.randn(1, N, H, D, device="cuda") for _ in range(3))
    probs = torch.empty(1, H, N, N, device="cuda")
    o = ops.attn_capture_fwd(q, k, v, probs, D ** -0.5)
    d_o = torch.randn(K, N, H, D, device="cuda") * 1e-2
    dprobs = torch.empty(K, H, N, N, device="cuda")
    us = kernel_time_us(lambda: ops.attn_capture_bwd(q, k, v, probs, d_o, dprobs, D ** -0.5, batch=K, o=o), 10,
                        torch.cuda.current_stream())
    flop = 4 * 2 * K * H * N * N * D
    kern = _mfma("attn_bwd_q_stream_kernel<32> + attn_bwd_kv_stream_kernel<32> (one encoder layer, K = 10 x 8 heads x 950^2)",
                 flop, us, FP32_MFMA_PEAK_TFLOPS,
                 "exact-fp32 MFMA; the dP slab (write once, read once: %.0f MB) is %.0f us at 8 TB/s"
                 % (2 * K * H * N * N * 4 / 1e6, 2 * K * H * N * N * 4 / 8e6))
    return {"workload": "BASELINE config 3 shape: DETR-R50 transformer + heads, 25x38 = 950 image tokens, 100 queries; K kept "
                        "queries of one image per pass (shared forward, batched backward, row-vector rules), fp32, hipGraph",
            "rate": round(20 / out[20] * 1e3, 1), "unit": "queries/s", "ms": round(out[20], 3),
            "K10": {"ms": round(out[10], 3), "ms_per_query": round(out[10] / 10, 3)},
            "K20": {"ms": round(out[20], 3), "ms_per_query": round(out[20] / 20, 3)},
            "evaluator": {"ms_per_image": round(ev_ms, 3), "images_per_s": round(1e3 / ev_ms, 1), "kept_queries_per_image": keep_top,
                          "queries_per_s": round(keep_top * 1e3 / ev_ms, 1),
                          "what": "forward + keep set + one K-slot pass (hipGraph, 8 slots) + Otsu masks per image, one device->host "
                                  "read per image, %d images after 2 warm-up images" % n_img},
            "lrp": {"ours_no_lrp_ms": round(lrp_no, 3), "ours_lrp_ms": round(lrp_yes, 3), "ratio": round(lrp_yes / lrp_no, 2),
                    "what": "Generator.generate_ours(img, [q]) per kept query, eager, per-query route (autograd backward): "
                            "use_lrp=False vs the default use_lrp=True (body relprop: closed-form rules + HIP attention-core kernels)"},
            "kernel": kern, "source": "measured in this run (HIP events); rocprofv3 kernel table of the same legs: profiles/r06_cfg_legs.txt"}


def leg_cfg4(kernel_time_us, reps=10):
    """cfg 4: LXMERT-base (9 language / 5 vision / 5 cross layers, 12 heads), T = 14 question tokens, I = 36 regions,
    batches of 32 samples per GPU: explain (GeneratorOurs) + the 9-step image perturbation test."""
    from transformer_mm_explainability_amd import lxmert_explainability as le
    from transformer_mm_explainability_amd import lxmert_model as lm
    from transformer_mm_explainability_amd import lxmert_perturbation as lp
    from transformer_mm_explainability_amd import ops
    torch.manual_seed(0)
    model = lm.LxmertForQuestionAnswering(lm.LxmertConfig()).cuda().eval()
    B, T, I, H = 32, 14, 36, 12
    gb = torch.Generator().manual_seed(2)
    batch = dict(input_ids=torch.randint(1, 30000, (B, T), generator=gb).cuda(), attention_mask=torch.ones(B, T).cuda(),
                 token_type_ids=torch.zeros(B, T, dtype=torch.long).cuda(),
                 visual_feats=torch.randn(B, I, 2048, generator=gb).cuda(), visual_pos=torch.rand(B, I, 4, generator=gb).cuda())
    run = le.GraphedGenerateOursBatch(model, batch)
    ms_explain = _timed_ms(lambda: run(batch), reps)
    pert = lp.LxmertPerturbation(model)
    cams = torch.rand(B, I, generator=gb).cuda()
    ms_pert_eager = _timed_ms(lambda: pert.perturbation_image(batch, cams), max(3, reps // 2))
    # what the evaluator runs per batch (examples/lxmert_perturbation_eval.py): cams + the 9-step test + accuracy replayed from ONE
    # hipGraph -- the eager form above is ~1100 launches and bound by the host
    R_t_t, R_t_i = torch.rand(B, T, T, generator=gb).cuda(), torch.rand(B, T, I, generator=gb).cuda()
    labels = torch.rand(B, model.config.num_qa_labels, generator=gb).cuda()
    pert_graph = lp.GraphedImagePerturbation(pert, batch, R_t_t, R_t_i, labels)
    ms_pert = _timed_ms(lambda: pert_graph(batch, R_t_t, R_t_i, labels), reps)
    del run, pert_graph
    # the LRP route (use_lrp=True, the generators' DEFAULT argument): one item per call, as perturbation.py:216-238 runs it
    import types
    item = {k: v[:1] for k, v in batch.items()}
    gen = le.GeneratorOurs(types.SimpleNamespace(model=model, text_len=T, image_boxes_len=I, forward=lambda it: model(**item)))
    lrp_no = _timed_ms(lambda: gen.generate_ours(None, use_lrp=False), 3)
    lrp_yes = _timed_ms(lambda: gen.generate_ours(None), 3)
    del gen
    # the one-launch rule schedule on slabs of the same sizes
    sm = lambda *s: torch.softmax(torch.randn(*s, device="cuda"), -1)
    gr = lambda *s: torch.randn(*s, device="cuda") * 1e-2
    pair = lambda nq, nk: (sm(B, H, nq, nk), gr(B, H, nq, nk))
    lang, vis = [pair(T, T) for _ in range(9)], [pair(I, I) for _ in range(5)]
    xlc, xic = [pair(T, I) for _ in range(5)], [pair(I, T) for _ in range(5)]
    xls, xis = [pair(T, T) for _ in range(5)], [pair(I, I) for _ in range(5)]
    nbytes = sum(a.numel() * 8 for grp in (lang, vis, xlc, xic, xls, xis) for a, _ in grp) + B * (T * T + T * I + I * I + I * T) * 4
    us = kernel_time_us(lambda: ops.lxmert_schedule(lang, vis, xlc, xic, xls, xis, check_diag="defer"), 20,
                        torch.cuda.current_stream())
    return {"workload": "BASELINE config 4 shape: LXMERT-base, T = 14 question tokens, 36 regions, batch 32 per GPU: explain "
                        "(GeneratorOurs, one hipGraph) + 9-step image perturbation test (two gathered step groups, one hipGraph), fp32",
            "rate": round(B / (ms_explain + ms_pert) * 1e3, 1), "unit": "samples/s", "ms": round(ms_explain + ms_pert, 3),
            "explain_ms": round(ms_explain, 3), "perturb_ms": round(ms_pert, 3), "perturb_eager_ms": round(ms_pert_eager, 3),
            "lrp": {"ours_no_lrp_ms": round(lrp_no, 3), "ours_lrp_ms": round(lrp_yes, 3), "ratio": round(lrp_yes / lrp_no, 2),
                    "what": "GeneratorOurs.generate_ours(item) per item, eager: use_lrp=False vs the default use_lrp=True"},
            "kernel": _hbm("lxmert_schedule_v2_kernel = mmx_lxmert_schedule (38 rule applications: chip-wide rule 5 + last-arriver schedule on the MFMA, "
                           "B = 32)", nbytes, us, "2 MB of slabs per sample; the serial 38-step schedule of a sample is the floor"),
            "source": "measured in this run (HIP events); rocprofv3 kernel table of the same legs: profiles/r06_cfg_legs.txt"}


def cfg5_step_flops(batch, executed=False):
    """Matrix FLOPs of one cfg-5 step (CLIP ViT-L/14@336 bf16 body: image tower 24 x 1024 x 16 heads x 577 tokens with a shared
    forward and row-relevancy backward, text tower 12 x 768 x 12 x 77), as ``step_flops`` counts them for cfg 2.
    ALGORITHMIC count: an attention backward is FOUR products (dP = dO.V^T, dV = P^T.dO, dQ = dS.K,
    dK = dS^T.Q: 8 N^2 d per head); ``executed=True`` counts the FIVE the two-kernel image-tower design runs (both kernels
    recompute dP)."""
    def tower(L, E, N, H, m_fwd, m_bwd, attn_products):
        gemm_fwd = L * 2 * m_fwd * 12 * E * E
        full, top, low = 2 * m_bwd * 12 * E * E, 2 * m_bwd * 3 * E * E + 2 * batch * 9 * E * E, 2 * m_bwd * 9 * E * E
        d = E // H
        attn = L * 4 * (m_fwd // N) * H * N * N * d + (L - 1) * attn_products * 2 * (m_bwd // N) * H * N * N * d \
            + 2 * (m_bwd // N) * H * N * N * d
        return gemm_fwd + (L - 2) * full + top + low, attn
    g_img, a_img = tower(24, 1024, 577, 16, 577, batch * 577, 5 if executed else 4)
    g_txt, a_txt = tower(12, 768, 77, 12, batch * 77, batch * 77, 4)
    return {"gemm": g_img + g_txt, "attention": a_img + a_txt, "total": g_img + g_txt + a_img + a_txt}


def cfg5_setup(batch, device, rank=0):
    """Model, inputs and the per-layer attention-backward launch of the cfg-5 leg (shared with ``bench.py --workload cfg5``)."""
    from transformer_mm_explainability_amd import clip_model, ops, tuned_gemms
    tuned_gemms.enable("clip_vitl14_336_bf16")      # pre-tuned selection for this body's GEMM shapes (tuning off)
    model = clip_model.random_init("ViT-L/14@336", seed=0).to(device)
    model.set_body_dtype(torch.bfloat16)
    g = torch.Generator().manual_seed(1 + rank)
    image = torch.randn(1, 3, 336, 336, generator=g).to(device)
    texts = torch.zeros(batch, 77, dtype=torch.long)
    g2 = torch.Generator().manual_seed(2 + rank)
    for b in range(batch):
        n = int(torch.randint(3, 11, (1,), generator=g2))
        texts[b, 0] = 49406
        texts[b, 1:1 + n] = torch.randint(1, 49405, (n,), generator=g2)
        texts[b, 1 + n] = 49407
    texts = texts.to(device)
    H, N, D = 16, 577, 64
    qkv = torch.randn(1, N, 3, H, D, device=device)
    d_o = (torch.randn(batch, N, H, D, device=device) * 1e-2).to(torch.bfloat16)
    probs = torch.empty(1, H, N, N, device=device, dtype=torch.bfloat16)
    o = ops.attn_capture_fwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], probs, D ** -0.5, mma_bf16=True)
    out = torch.empty(batch, N, 3, H, D, device=device, dtype=torch.bfloat16)
    rel = torch.zeros(batch, N, device=device)
    rel[:, 0] = 1

    def attn_layer():
        ops.attn_capture_bwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], probs, d_o, None, D ** -0.5, batch=batch, o=o,
                             out=(out[:, :, 0], out[:, :, 1], out[:, :, 2]), mma_bf16=True, rel_row=rel)
    # FLOPs per launch pair: ALGORITHMIC = 4 products of 2 N^2 d each per (sample, head); the pair EXECUTES 5 (dP in both kernels)
    return model, image, texts, attn_layer, {"algorithmic": 4 * 2 * batch * H * N * N * D, "executed": 5 * 2 * batch * H * N * N * D}


def leg_cfg5(kernel_time_us, reps=3, batch=128):
    """cfg 5 on one GPU: CLIP ViT-L/14@336 (577 image tokens), bf16 body, 128 pairs per GPU, all 24 + 12 layers, eager."""
    from transformer_mm_explainability_amd import clip_explainability as ce
    model, image, texts, attn_layer, attn_flops = cfg5_setup(batch, torch.device("cuda"))
    torch.cuda.reset_peak_memory_stats()
    ms = _timed_ms(lambda: ce.interpret(image, texts, model, "cuda", start_layer=0, start_layer_text=0), reps, warm=2)
    # variant (opt-in, exact): the captions are 5-12 tokens of the 77-token context; under the causal mask the positions behind a
    # caption's EOT token cannot reach it, so the text tower may run on the longest caption only (trim_text_padding)
    ms_trim = _timed_ms(lambda: ce.interpret(image, texts, model, "cuda", start_layer=0, start_layer_text=0,
                                             trim_text_padding=True), reps, warm=2)
    us = kernel_time_us(attn_layer, 5, torch.cuda.current_stream())
    fl, fl_exec = cfg5_step_flops(batch), cfg5_step_flops(batch, executed=True)
    step_tf = fl["total"] / (ms * 1e-3) / 1e12
    kernel = _mfma("attn_bwd_q_v3_kernel + attn_bwd_kv_v4_kernel (+ the two prep kernels; one image-tower layer, B = 128, "
                   "row-relevancy mode, attention_bf16_v3.hip)", attn_flops["algorithmic"], us, BF16_MFMA_PEAK_TFLOPS,
                   "numerator = ALGORITHMIC FLOPs (4 products); the pair executes 5 (dP in both kernels: the deterministic two-kernel "
                   "split, no atomics on dQ): executed_* beside it.  Bound by VALU issue of the elementwise work between the products "
                   "(relevancy partial, dS, bf16 packing: ~12 VALU instructions per 32-cycle MFMA), not by the matrix cores: "
                   "profiles/r06_cfg5_probe.txt")
    kernel["executed_flop_per_launch"] = int(attn_flops["executed"])
    kernel["executed_achieved"] = round(attn_flops["executed"] / us / 1e6, 1)
    kernel["executed_frac"] = round(attn_flops["executed"] / us / 1e6 / BF16_MFMA_PEAK_TFLOPS, 4)
    slab = _cfg5_slab_chain_variant(model, image, texts, batch, kernel_time_us)
    return {"roofline_step": {"bound": "mfma", "achieved": round(step_tf, 1), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": round(step_tf / BF16_MFMA_PEAK_TFLOPS, 4), "flop_per_step": fl,
                              "executed_flop_per_step": fl_exec["total"],
                              "executed_frac": round(fl_exec["total"] / (ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4),
                              "what": "ALGORITHMIC bf16 matrix FLOPs of the step (body GEMMs forward + input-gradient, attention "
                                      "products: 4 per backward) / ms; executed_* counts the 5th product the two-kernel backward re-does"},
            "workload": "BASELINE config 5 shape: CLIP ViT-L/14@336 (577 image tokens), batch 128 per GPU, all 24+12 layers, "
                        "bf16 body (fp32 accumulation / LayerNorm / softmax / relevancy), row-relevancy image tower, eager",
            "rate": round(batch / ms * 1e3, 1), "unit": "maps/s", "ms": round(ms, 3),
            "resident_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
            "kernel": kernel,
            "variant_slab_chain": slab,
            "variant_trim_text_padding": {"ms": round(ms_trim, 3), "rate": round(batch / ms_trim * 1e3, 1),
                                          "note": "same maps (exact); NOT the headline of this leg: the reference runs all 77 positions"},
            "source": "measured in this run (HIP events); rocprofv3 kernel table of the same legs: profiles/r06_cfg_legs.txt"}


def _cfg5_slab_chain_variant(model, image, texts, batch, kernel_time_us):
    """cfg 5 on the route SURVEY section 8(d) prices (VERDICT r04 missing #4): gradient slabs captured for all 24 image-tower layers
    (bf16) and the FULL-MATRIX chain R <- R + A_bar . R at N = 577 (avg_heads + exact-fp32 bmm per layer: N > 128) instead of the
    row-relevancy backward.  Reports the step and, stand-alone with HIP events, the chain segment as GB/s of the slabs it reads."""
    from transformer_mm_explainability_amd import clip_explainability as ce
    vis = model.visual
    try:
        vis.row_relevancy_ok = lambda: False
        f = lambda: ce.interpret(image, texts, model, "cuda", start_layer=0, start_layer_text=0)   # noqa: E731
        ms = _timed_ms(f, 2, warm=1)
        tr = vis.transformer
        b = tr.buffers
        L, H, N = tr.layers, tr.heads, b.probs[0].shape[-1]
        plan = ce._plan(b, 0, L, batch, b.shared_probs and batch > 1, getattr(tr, "half_chain", False))
        us = kernel_time_us(plan.launch, 3, torch.cuda.current_stream())
        esz = b.grads[0].element_size()
        shared = bool(b.shared_probs)
        read = L * (batch * H * N * N * esz + (1 if shared else batch) * H * N * N * b.probs[0].element_size())
        out = {"ms": round(ms, 3), "rate": round(batch / ms * 1e3, 1), "unit": "maps/s",
               "chain_segment": {"what": "24 x (avg_heads_kernel<bf16> + bmm_f32_kernel 577^3) for the image tower, B = %d, stand-alone" % batch,
                                 "us": round(us, 1), "slab_bytes_read": int(read), "achieved": round(read / us / 1e3, 1), "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": round(read / us / 1e3 / HBM_PEAK_GBS, 4),
                                 "flop": int(L * batch * 2 * N * N * N), "mfma_tflops": round(L * batch * 2 * N * N * N / us / 1e6, 1),
                                 "note": "the probability slab is ONE forward shared by the batch here (%s): bytes = every gradient slab + the "
                                         "shared probabilities once per layer; SURVEY 8(d)'s 514.8 MB/map prices per-sample slabs of both.  "
                                         "At 577 tokens the chain is fp32-MFMA-bound (2 N^3 per layer and sample), not HBM-bound"
                                         % ("shared" if shared else "per sample")},
               "note": "same maps as the row-relevancy leg within the bf16 slab rounding (tests/test_gpu_parity_fullsize.py::test_cfg5_bf16_body_vs_oracle)"}
    except Exception as exc:                                   # a variant must not take the leg down
        out = {"error": "%s: %s" % (type(exc).__name__, exc)}
    finally:
        if "row_relevancy_ok" in vis.__dict__:
            del vis.row_relevancy_ok
    return out


LEGS = (("cfg1", leg_cfg1), ("cfg3", leg_cfg3), ("cfg4", leg_cfg4), ("cfg5", leg_cfg5))


def run_all(kernel_time_us, log=lambda m: None, only=None):
    import gc
    out = {}
    for name, fn in LEGS:
        if only and name not in only:
            continue
        t0 = time.perf_counter()
        try:
            out[name] = fn(kernel_time_us)
        except Exception as exc:                                  # a leg must never take the headline line down
            out[name] = {"error": "%s: %s" % (type(exc).__name__, exc), "trace": traceback.format_exc()[-600:]}
        out[name]["leg_seconds"] = round(time.perf_counter() - t0, 1)
        log("%s leg done in %.1f s" % (name, time.perf_counter() - t0))
        gc.collect()
        torch.cuda.empty_cache()
    return out
