"""-m gpu: end-to-end CLIP path (HIP attention-capture op inside the PyTorch body + fused chain kernel)
against (a) the reference's own outputs on a tiny CLIP (golden fixture) and (b) the torch CPU oracle at
ViT-B/32 shapes.  Tolerance 1e-5 abs on relevancy maps (north_star), fp32."""
import json

import pytest
import torch

pytestmark = pytest.mark.gpu


from parity import close  # noqa: E402  (atol only: rtol = 0)


def load_tiny(golden):
    from transformer_mm_explainability_amd import clip_model
    g = golden("clip_tiny")
    cfg = json.loads(str(g["cfg_json"]))
    model = clip_model.CLIP(**cfg).float().eval()
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("w__")}
    model.load_state_dict(sd)      # the reference's own parameter names load unchanged
    return g, model.cuda()


@pytest.mark.parametrize("share", [True, False])
@pytest.mark.parametrize("tag,sl,slt", [("last", -1, -1), ("all", 0, 0), ("mid", 1, 2)])
def test_interpret_matches_reference_outputs(golden, tag, sl, slt, share):
    """share=True: image-tower forward run once + batched hand-written backward; False: B copies like the reference."""
    from transformer_mm_explainability_amd import clip_explainability as ce
    g, model = load_tiny(golden)
    image, texts = torch.from_numpy(g["image"]).cuda(), torch.from_numpy(g["texts"]).cuda()
    R_text, R_image = ce.interpret(image, texts, model, "cuda", start_layer=sl, start_layer_text=slt,
                                   share_image_forward=share)
    close(R_text, g["R_text_" + tag])
    close(R_image, g["R_image_" + tag])
    assert all(p.requires_grad for p in model.parameters())          # _Frozen restored
    assert all(p.grad is None for p in model.parameters())           # no weight gradients were computed


@pytest.mark.parametrize("share", [True, False])
def test_capture_slabs_match_reference_hooks(golden, share):
    """probs/grads slabs == what the reference's set_attn_probs / set_attn_grad hooks saw (in shared-forward mode the
    image tower keeps ONE copy of the identical probabilities, the gradients stay per sample)."""
    from transformer_mm_explainability_amd import clip_explainability as ce
    g, model = load_tiny(golden)
    image, texts = torch.from_numpy(g["image"]).cuda(), torch.from_numpy(g["texts"]).cuda()
    ce.interpret(image, texts, model, "cuda", 0, 0, share_image_forward=share)
    vis = list(model.visual.transformer.resblocks.children())
    txt = list(model.transformer.resblocks.children())
    # tolerances of DESIGN.md section 6: probabilities 2e-6, gradients 2e-5 (here 5e-6: the values are O(0.1))
    for l, blk in enumerate(vis):
        want = g["img_attn"][l]
        close(blk.attn_probs, want[:blk.attn_probs.shape[0]] if share else want, atol=2e-6, rtol=1e-4, what="intermediate")
        close(blk.attn_grad, g["img_grad"][l], atol=5e-6, rtol=1e-4, what="intermediate")
    for l, blk in enumerate(txt):
        close(blk.attn_probs, g["txt_attn"][l], atol=2e-6, rtol=1e-4, what="intermediate")
        close(blk.attn_grad, g["txt_grad"][l], atol=5e-6, rtol=1e-4, what="intermediate")
    with torch.no_grad():
        pass
    logits, _ = model(image.repeat(texts.shape[0], 1, 1, 1), texts)
    close(logits, g["logits_per_image"], atol=2e-5, rtol=1e-4, what="intermediate")


@pytest.mark.parametrize("share", [True, False])
@pytest.mark.parametrize("tag,sl,slt", [("last", -1, -1), ("all", 0, 0), ("mid", 1, 2)])
def test_fp16_mode_matches_the_references_fp16_model(golden, tag, sl, slt, share):
    """``set_body_dtype(torch.float16)`` -- the reference's OWN half-precision mode -- against the reference's model after
    ``convert_weights`` (CLIP/clip/model.py:381-402) + notebook cell 6 with its fp16 R chain (clip_tiny_fp16.npz, run on the
    CPU).  Ours rounds the same weights to fp16 and applies the chain's fp16 roundings, but keeps the residual stream, LayerNorm
    and softmax in fp32, so it sits between the reference's fp16 and fp32 results: the bar is the distance between those two
    (4e-3 of the largest entry; measured 6.5e-4 for R_text and 1.7e-3 for the image relevance on this fixture)."""
    from transformer_mm_explainability_amd import clip_explainability as ce
    g, model = load_tiny(golden)
    g16 = golden("clip_tiny_fp16")
    image, texts = torch.from_numpy(g["image"]).cuda(), torch.from_numpy(g["texts"]).cuda()
    model.set_body_dtype(torch.float16)
    R_text, R_image = ce.interpret(image, texts, model, "cuda", start_layer=sl, start_layer_text=slt, share_image_forward=share)
    assert R_text.dtype == torch.float16 and R_image.dtype == torch.float16          # like the reference's returns
    for got, name in ((R_text, "R_text_" + tag), (R_image, "R_image_" + tag)):
        want16, want32 = torch.from_numpy(g16[name]).float(), torch.from_numpy(g[name])
        top = float(want32.abs().max())
        err16 = float((got.float().cpu() - want16).abs().max())
        err32 = float((got.float().cpu() - want32).abs().max())
        assert err16 <= 4e-3 * top and err32 <= 4e-3 * top, (name, err16, err32, top)
    # and back: the exact path is untouched by the excursion
    model.set_body_dtype(torch.float32)
    R_text, R_image = ce.interpret(image, texts, model, "cuda", start_layer=sl, start_layer_text=slt, share_image_forward=share)
    assert R_text.dtype == torch.float32
    close(R_text, g["R_text_" + tag])
    close(R_image, g["R_image_" + tag])


@pytest.mark.parametrize("tag,sl,slt", [("last", -1, -1), ("all", 0, 0), ("mid", 1, 2)])
def test_interpret_trim_text_padding(golden, tag, sl, slt):
    """Running the text tower only up to the last EOT token gives the reference's full [B, 77, 77] result."""
    from transformer_mm_explainability_amd import clip_explainability as ce
    g, model = load_tiny(golden)
    image, texts = torch.from_numpy(g["image"]).cuda(), torch.from_numpy(g["texts"]).cuda()
    assert int(texts.argmax(-1).max()) + 1 < texts.shape[1]          # the fixture's captions are padded
    R_text, R_image = ce.interpret(image, texts, model, "cuda", start_layer=sl, start_layer_text=slt,
                                   trim_text_padding=True)
    close(R_text, g["R_text_" + tag])
    close(R_image, g["R_image_" + tag])


def test_full_backward_param_grads_match_oracle(golden):
    """With capture_only off the op is a normal differentiable attention: parameter grads == torch CPU oracle."""
    from oracle import clip_torch
    g, model = load_tiny(golden)
    cfg = json.loads(str(g["cfg_json"]))
    image, texts = torch.from_numpy(g["image"]), torch.from_numpy(g["texts"])
    B = texts.shape[0]
    logits, _ = model(image.cuda().repeat(B, 1, 1, 1), texts.cuda())
    model.zero_grad()
    logits.diagonal().sum().backward()
    sd = clip_torch.prepare_state_dict({k: v.cpu() for k, v in model.state_dict().items()}, cfg["transformer_heads"])
    ref_logits, _, _ = clip_torch.forward(sd, image.repeat(B, 1, 1, 1), texts)
    ref_logits.diagonal().sum().backward()
    for name, p in model.named_parameters():
        if name == "logit_scale":
            continue
        close(p.grad, sd[name].grad, atol=2e-5, rtol=1e-3)


def test_vit_b32_shapes_vs_oracle():
    """CLIP ViT-B/32 architecture (random init), B=2: maps within 1e-5 of the reference-style CPU path."""
    from oracle import clip_torch
    from transformer_mm_explainability_amd import clip_explainability as ce
    from transformer_mm_explainability_amd import clip_model
    model = clip_model.random_init("ViT-B/32", seed=0)
    g = torch.Generator().manual_seed(1)
    image = torch.randn(1, 3, 224, 224, generator=g)
    texts = torch.zeros(2, 77, dtype=torch.long)
    g2 = torch.Generator().manual_seed(2)
    for b in range(2):
        n = 4 + 3 * b
        texts[b, 0] = 49406
        texts[b, 1:1 + n] = torch.randint(1, 49405, (n,), generator=g2)
        texts[b, 1 + n] = 49407
    sd = clip_torch.prepare_state_dict(model.state_dict(), 8)
    want_text, want_img = clip_torch.interpret(sd, image, texts, 0, 0)
    model = model.cuda()
    for share in (True, False):
        R_text, R_image = ce.interpret(image.cuda(), texts.cuda(), model, "cuda", 0, 0, share_image_forward=share)
        close(R_text, want_text.numpy())
        close(R_image, want_img.numpy())
    want_text, want_img = clip_torch.interpret(sd, image, texts)         # notebook default: last layer only
    R_text, R_image = ce.interpret(image.cuda(), texts.cuda(), model, "cuda")
    close(R_text, want_text.numpy())
    close(R_image, want_img.numpy())
    single = ce.interpret_single(image.cuda(), texts.cuda(), model, "cuda", index=1)
    assert single.shape == (49,)


@pytest.mark.parametrize("trim", [False, True])
def test_graphed_interpret_matches_eager(golden, trim):
    """hipGraph replay of the whole step == eager interpret, also after new inputs are copied in."""
    from transformer_mm_explainability_amd import clip_explainability as ce
    g, model = load_tiny(golden)
    image, texts = torch.from_numpy(g["image"]).cuda(), torch.from_numpy(g["texts"]).cuda()
    run = ce.GraphedInterpret(model, image, texts, 0, 0, trim_text_padding=trim)
    R_text, R_image = run()
    close(R_text, g["R_text_all"])
    close(R_image, g["R_image_all"])
    image2 = image.flip(-1).contiguous()
    texts2 = texts.roll(1, 0).contiguous()
    want_t, want_i = ce.interpret(image2, texts2, model, "cuda", 0, 0)
    got_t, got_i = run(image2, texts2)
    close(got_t, want_t.cpu().numpy(), atol=2e-6, rtol=1e-4, what="intermediate")
    close(got_i, want_i.cpu().numpy(), atol=2e-6, rtol=1e-4, what="intermediate")


def test_full_size_batch_properties():
    """BASELINE.json's measured configuration (ViT-B/32, batch 64, all layers) through size-independent properties:
    every sample of the full batch equals the same sample explained in a batch of 2 (samples are independent), the text
    relevancy is causal (lower triangular), >= identity on the diagonal and exactly the identity beyond the prompt's
    EOT token, image relevancies are non-negative, and the hipGraph replay returns the eager values."""
    from transformer_mm_explainability_amd import clip_explainability as ce
    from transformer_mm_explainability_amd import clip_model
    model = clip_model.random_init("ViT-B/32", seed=0).cuda()
    g = torch.Generator().manual_seed(1)                                 # bench.py's synthetic workload
    image = torch.randn(1, 3, 224, 224, generator=g).cuda()
    texts = torch.zeros(64, 77, dtype=torch.long)
    g2 = torch.Generator().manual_seed(2)
    for b in range(64):
        n = int(torch.randint(3, 11, (1,), generator=g2))
        texts[b, 0] = 49406
        texts[b, 1:1 + n] = torch.randint(1, 49405, (n,), generator=g2)
        texts[b, 1 + n] = 49407
    texts = texts.cuda()
    R_text, R_image = (t.clone() for t in ce.interpret(image, texts, model, "cuda", 0, 0))
    assert R_text.shape == (64, 77, 77) and R_image.shape == (64, 49)
    assert torch.isfinite(R_text).all() and torch.isfinite(R_image).all() and (R_image >= 0).all()
    for pick in ([0, 1], [37, 63]):
        sub_text, sub_image = ce.interpret(image, texts[pick], model, "cuda", 0, 0)
        torch.testing.assert_close(sub_text, R_text[pick], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(sub_image, R_image[pick], rtol=1e-4, atol=1e-5)
    assert (R_text.triu(1) == 0).all()                                   # causal attention: no relevancy from the future
    assert (R_text.diagonal(dim1=1, dim2=2) >= 1).all()
    eot = texts.argmax(dim=-1)
    eye = torch.eye(77, device="cuda")
    for b in (0, 5, 63):
        n = int(eot[b]) + 1
        assert torch.equal(R_text[b, n:, :], eye[n:, :])                 # padding rows: zero gradient -> identity
    run = ce.GraphedInterpret(model, image, texts, start_layer=0, start_layer_text=0)
    g_text, g_image = run()
    torch.testing.assert_close(g_text, R_text, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(g_image, R_image, rtol=1e-5, atol=1e-6)


def test_bf16_backward_gemms_stay_close(golden):
    """``backward_gemm_dtype = bfloat16`` on the image tower (config-5 style low-precision mode): only the input-gradient
    GEMMs of the hand-written backward run on the bf16 MFMA; the maps stay within bf16 precision of the reference's."""
    from transformer_mm_explainability_amd import clip_explainability as ce
    g, model = load_tiny(golden)
    image, texts = torch.from_numpy(g["image"]).cuda(), torch.from_numpy(g["texts"]).cuda()
    model.visual.transformer.backward_gemm_dtype = torch.bfloat16
    R_text, R_image = ce.interpret(image, texts, model, "cuda", start_layer=0, start_layer_text=0)
    close(R_text, g["R_text_all"])                                    # the text tower is untouched
    want = torch.from_numpy(g["R_image_all"]).cuda()
    assert (R_image - want).abs().max() <= 2e-2 * want.abs().max()
    assert torch.nn.functional.cosine_similarity(R_image, want, dim=-1).min() > 0.9995
    model.visual.transformer.backward_gemm_dtype = torch.float32
    close(ce.interpret(image, texts, model, "cuda", start_layer=0, start_layer_text=0)[1], g["R_image_all"])


def test_graphed_interpret_survives_buffer_replacement(golden):
    """a captured graph holds raw slab addresses.  An eager call with another batch / sharing mode
    makes the towers allocate NEW slabs; the graph must keep its own alive and still replay correctly afterwards, and
    ``blk.attn_probs`` must again point at what the replay wrote."""
    from transformer_mm_explainability_amd import clip_explainability as ce
    g, model = load_tiny(golden)
    image, texts = torch.from_numpy(g["image"]).cuda(), torch.from_numpy(g["texts"]).cuda()
    run = ce.GraphedInterpret(model, image, texts, 0, 0)
    pinned = model.visual.transformer.buffers
    ce.interpret(image, texts[:2], model, "cuda", 0, 0)                           # other batch -> new slabs
    ce.interpret(image, texts, model, "cuda", 0, 0, share_image_forward=False)    # other sharing mode -> new slabs
    assert model.visual.transformer.buffers is not pinned
    junk = [torch.randn(1 << 20, device="cuda") for _ in range(8)]                # would land on freed slabs
    R_text, R_image = run()
    close(R_text, g["R_text_all"])
    close(R_image, g["R_image_all"])
    assert model.visual.transformer.buffers is pinned
    blk = model.visual.transformer.resblocks[0]
    close(blk.attn_grad, g["img_grad"][0], atol=1e-6, rtol=1e-4, what="intermediate")
    del junk
