"""Host logic of the batched perturbation evaluator (runs on CPU): which regions / tokens each of the 9 steps keeps
equals the reference loop's own selection (lxmert/lxmert/perturbation.py:110-117 and :158-172, restated inline)."""
import torch

from transformer_mm_explainability_amd import lxmert_perturbation as lp


def test_image_keep_masks_match_reference_selection():
    g = torch.Generator().manual_seed(0)
    cam = torch.rand(36, generator=g)
    for positive in (False, True):
        keep = lp.image_keep_masks(cam, is_positive_pert=positive)
        assert keep.shape == (9, 36)
        c = -cam if positive else cam
        for s, step in enumerate(lp.PERT_STEPS):
            k = int((1 - step) * 36)
            want = torch.zeros(36)
            want[c.topk(k=k, dim=-1).indices] = 1
            assert torch.equal(keep[s], want)
        assert keep[0].sum() == 36 and keep[-1].sum() == 0


def test_text_keep_batch_matches_reference_gather():
    g = torch.Generator().manual_seed(1)
    T = 14
    ids = torch.randint(5, 1000, (1, T), generator=g)
    types = torch.zeros(1, T, dtype=torch.long)
    cam = torch.rand(T, generator=g)
    for positive in (False, True):
        got_ids, got_types, mask = lp.text_keep_batch(ids, types, cam, is_positive_pert=positive)
        c = -cam if positive else cam
        for s, step in enumerate(lp.PERT_STEPS):
            pure = c[1:-1]
            k = int((1 - step) * pure.shape[0])
            top = pure.topk(k=k, dim=-1).indices.tolist()
            kept = sorted([0, T - 1] + [i + 1 for i in top])                  # perturbation.py:166-170
            n = len(kept)
            assert mask[s].tolist() == [1.0] * n + [0.0] * (T - n)
            assert got_ids[s, :n].tolist() == ids[0, kept].tolist()
            assert got_ids[s, n:].abs().sum() == 0 and got_types[s].abs().sum() == 0
        assert mask[-1].sum() == 2                                            # only [CLS] and [SEP] survive step 1.0


def test_normalize_cams_and_accuracy():
    R_t_t = torch.tensor([[1.0, 3.0, 2.0], [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]])
    R_t_i = torch.tensor([[2.0, 6.0], [0.0, 0.0], [0.0, 0.0]])
    cam_image, cam_text = lp.normalize_cams(R_t_t, R_t_i)
    assert cam_image.tolist() == [0.0, 1.0] and cam_text.tolist() == [0.0, 1.0, 0.5]
    scores = torch.tensor([[0.1, 0.9, 0.0], [0.7, 0.2, 0.1]])
    labels = torch.tensor([0.3, 1.0, 0.0])
    assert lp.LxmertPerturbation.accuracy(scores, labels).tolist() == [1.0, 0.30000001192092896]


def test_visualbert_text_keep_batch_matches_reference_gather():
    """evaluation_loop.py:128-150 restated: keep [CLS], the inner top-k, the '?' token at cls_index and [SEP]."""
    from transformer_mm_explainability_amd import visualbert_perturbation as vp
    g = torch.Generator().manual_seed(3)
    n_text = 11
    ids = torch.randint(5, 1000, (1, n_text), generator=g)
    seg = torch.zeros(1, n_text, dtype=torch.long)
    cls_index = n_text - 2
    scores = torch.rand(cls_index - 1, generator=g)                      # method_cam[0, 1:cls_index]
    got_ids, got_seg, mask = vp.text_keep_batch(ids, seg, scores, n_text)
    for s, step in enumerate(vp.PERT_STEPS):
        k = int((1 - step) * len(scores))
        top = scores.topk(k=k, dim=-1).indices.tolist()
        kept = sorted([0, cls_index, cls_index + 1] + [i + 1 for i in top])
        n = len(kept)
        assert mask[s].tolist() == [1] * n + [0] * (n_text - n)
        assert got_ids[s, :n].tolist() == ids[0, kept].tolist()
        assert int(mask[s].sum()) - 2 == kept.index(cls_index)            # the 'vqa' pooler still reads the '?' token


def test_length_buckets():
    from transformer_mm_explainability_amd import sharding
    lengths = [7, 9, 7, 7, 12, 9, 7, 7]
    got = sharding.length_buckets(lengths, max_batch=3)
    assert got == [(7, [0, 2, 3]), (7, [6, 7]), (9, [1, 5]), (12, [4])]
    assert sorted(p for _, ps in got for p in ps) == list(range(len(lengths)))


def test_normalize_cams_batch_matches_per_item():
    g = torch.Generator().manual_seed(4)
    R_t_t, R_t_i = torch.rand(3, 5, 5, generator=g), torch.rand(3, 5, 8, generator=g)
    ci, ct = lp.normalize_cams_batch(R_t_t, R_t_i)
    for b in range(3):
        one_i, one_t = lp.normalize_cams(R_t_t[b], R_t_i[b])
        assert torch.allclose(ci[b], one_i) and torch.allclose(ct[b], one_t)


def test_partial_scores_resume(tmp_path):
    """Evaluator resume: rows survive a restart, a torn last line is ignored, table() follows the requested order."""
    from transformer_mm_explainability_amd import sharding
    store = sharding.PartialScores(str(tmp_path), rank=3)
    store.add([11, 4], torch.tensor([[1.0, 0.5], [0.0, 0.25]]))
    store.add([7], torch.tensor([[0.75, 0.125]]))
    store.close()
    with open(store.path, "a") as f:
        f.write('{"ids": [99], "rows": [[0.')                       # killed mid-write
    again = sharding.PartialScores(str(tmp_path), rank=3)
    assert again.done() == {11, 4, 7}
    assert again.table([7, 11, 4]).tolist() == [[0.75, 0.125], [1.0, 0.5], [0.0, 0.25]]
    again.add([99], torch.tensor([[1.0, 1.0]]))
    again.close()
    assert sharding.PartialScores(str(tmp_path), rank=3).done() == {11, 4, 7, 99}
    assert sharding.PartialScores(str(tmp_path), rank=0).done() == set()


def test_forward_for_backward_freezes_when_possible_and_falls_back_otherwise():
    """rules.forward_for_backward: parameters frozen if the score stays differentiable (input leaf / capture-op bodies),
    else the reference route (parameters requiring grad); requires_grad flags are always restored."""
    import torch.nn as nn
    from transformer_mm_explainability_amd import rules
    lin = nn.Linear(4, 3)
    x_leaf = torch.randn(2, 4, requires_grad=True)
    out = rules.forward_for_backward(lin, lambda: lin(x_leaf))
    out.sum().backward()
    assert lin.weight.grad is None and x_leaf.grad is not None          # frozen route: activation gradients only
    assert all(p.requires_grad for p in lin.parameters())
    x_const = torch.randn(2, 4)
    out = rules.forward_for_backward(lin, lambda: lin(x_const))          # detached under frozen parameters -> fallback
    out.sum().backward()
    assert lin.weight.grad is not None
    with rules.frozen_parameters(lin):
        assert not any(p.requires_grad for p in lin.parameters())
    assert all(p.requires_grad for p in lin.parameters())
    not_a_module = type("M", (), {})()
    assert rules.forward_for_backward(not_a_module, lambda: x_leaf * 2).requires_grad


def test_backward_gemm_cache_follows_the_parameter():
    """ops.backward_gemm: fp32 = plain matmul; bf16 = converted weight cached per parameter, refreshed after an in-place
    update, dropped with the parameter."""
    import gc
    from transformer_mm_explainability_amd import ops
    w = torch.nn.Parameter(torch.randn(8, 16))
    x = torch.randn(3, 8)
    assert torch.equal(ops.backward_gemm(x, w), x @ w)
    y = ops.backward_gemm(x, w, torch.bfloat16)
    assert y.dtype == torch.float32 and torch.allclose(y, x @ w, rtol=3e-2, atol=3e-2)
    cached = ops._GEMM_WEIGHTS[id(w)][torch.bfloat16][1]
    assert ops.backward_gemm(x, w, torch.bfloat16) is not None and ops._GEMM_WEIGHTS[id(w)][torch.bfloat16][1] is cached
    with torch.no_grad():
        w.mul_(2.0)
    y2 = ops.backward_gemm(x, w, torch.bfloat16)
    assert ops._GEMM_WEIGHTS[id(w)][torch.bfloat16][1] is not cached and torch.allclose(y2, 2 * y, rtol=1e-2, atol=1e-2)
    n = len(ops._GEMM_WEIGHTS)
    del w, cached
    gc.collect()
    assert len(ops._GEMM_WEIGHTS) == n - 1


def test_bf16_body_gemm_helpers():
    """The GEMM helpers of the bf16 body (``CLIP.set_body_dtype``): ``ops.linear`` rounds activations and the (cached)
    weight to bf16 and returns fp32 -- through ``aten::mm.dtype`` where the build has it, else by widening the bf16 result;
    ``backward_gemm_bf16`` keeps the gradient stream in bf16; fp32 stays the plain library call."""
    from transformer_mm_explainability_amd import ops
    lin = torch.nn.Linear(32, 24)
    x = torch.randn(5, 7, 32)
    assert torch.equal(ops.linear(x, lin.weight, lin.bias), torch.nn.functional.linear(x, lin.weight, lin.bias))
    y = ops.linear(x, lin.weight, lin.bias, torch.bfloat16)
    want = torch.nn.functional.linear(x.bfloat16().float(), lin.weight.detach().bfloat16().float(), lin.bias)
    assert y.dtype == torch.float32 and y.shape == (5, 7, 24)
    assert torch.allclose(y, want, rtol=2e-2, atol=2e-2)           # (bf16-rounded result on builds without mm.dtype)
    g = torch.randn(5, 7, 24).bfloat16()
    d = ops.backward_gemm_bf16(g, lin.weight)
    assert d.dtype == torch.bfloat16 and d.shape == (5, 7, 32)
    assert torch.allclose(d.float(), g.float() @ lin.weight.detach().bfloat16().float(), rtol=3e-2, atol=3e-2)
    assert ops._GEMM_WEIGHTS[id(lin.weight)][torch.bfloat16][1].dtype == torch.bfloat16     # one cached copy serves both


def test_clip_body_dtype_switch():
    """``CLIP.set_body_dtype``: image tower -> bf16 slabs + GEMMs + attention products, text tower -> bf16 GEMMs only (its
    77-token attention stays on the exact-fp32 register-resident kernels); back to fp32 restores everything."""
    from transformer_mm_explainability_amd import clip_model
    m = clip_model.CLIP(32, 32, 2, 128, 8, 12, 64, 64, 2, 2)
    vis, txt = m.visual.transformer, m.transformer
    assert (vis.capture_dtype, vis.forward_gemm_dtype, vis.attention_mma_bf16) == (torch.float32, torch.float32, False)
    m.set_body_dtype(torch.bfloat16)
    assert (vis.capture_dtype, vis.forward_gemm_dtype, vis.backward_gemm_dtype, vis.attention_mma_bf16) == \
        (torch.bfloat16, torch.bfloat16, torch.bfloat16, True)
    assert (txt.capture_dtype, txt.forward_gemm_dtype, txt.backward_gemm_dtype) == (torch.float32, torch.bfloat16, torch.bfloat16)
    assert not m.visual.row_relevancy_ok()          # 17 tokens: the whole-head kernels, no row mode
    m.set_body_dtype(torch.float32)
    assert (vis.capture_dtype, vis.forward_gemm_dtype, vis.attention_mma_bf16, txt.forward_gemm_dtype) == \
        (torch.float32, torch.float32, False, torch.float32)
    # the reference's own half-precision mode (convert_weights): fp16 GEMMs, fp16 long-sequence slabs, the fp16 relevancy chain
    m.set_body_dtype(torch.float16)
    assert (vis.capture_dtype, vis.forward_gemm_dtype, vis.backward_gemm_dtype, vis.attention_mma_bf16, vis.half_chain) == \
        (torch.float16, torch.float16, torch.float16, False, True)
    assert (txt.capture_dtype, txt.forward_gemm_dtype, txt.half_chain) == (torch.float32, torch.float16, True)
    m.set_body_dtype(torch.float32)
    assert (vis.capture_dtype, vis.forward_gemm_dtype, vis.half_chain, txt.half_chain) == (torch.float32, torch.float32, False, False)
    import pytest
    with pytest.raises(ValueError):
        m.set_body_dtype(torch.float64)


def test_batched_keep_builders_equal_the_per_sample_ones():
    """``image_keep_masks`` on ``[B, I]`` and ``text_keep_batches`` on a padded batch of ragged questions (a few launches for the
    whole batch) against the one-sample calls pinned above -- including tied scores (stable ranking: lower index first)."""
    g = torch.Generator().manual_seed(3)
    cams = torch.rand(5, 36, generator=g)
    cams[1, 7] = cams[1, 20] = cams[1, 3]                                   # ties
    cams[2] = 0.5                                                            # all equal
    for positive in (False, True):
        got = lp.image_keep_masks(cams, is_positive_pert=positive)
        assert got.shape == (5, 9, 36)
        for b in range(5):
            assert torch.equal(got[b], lp.image_keep_masks(cams[b], is_positive_pert=positive))
    P, lens = 20, [20, 14, 6, 3, 9]
    ids = torch.randint(5, 1000, (5, P), generator=g)
    types = torch.randint(0, 2, (5, P), generator=g)
    cam_t = torch.rand(5, P, generator=g)
    cam_t[4, 2] = cam_t[4, 5] = cam_t[4, 3]
    for b, n in enumerate(lens):
        ids[b, n:] = 0
        types[b, n:] = 0
        cam_t[b, n:] = 0
    for positive in (False, True):
        got = lp.text_keep_batches(ids, types, cam_t, is_positive_pert=positive, n_tokens=lens)
        for b, n in enumerate(lens):
            want = lp.text_keep_batch(ids[b:b + 1], types[b:b + 1], cam_t[b], is_positive_pert=positive, n_tokens=n)
            for k in range(3):
                assert torch.equal(got[k][9 * b: 9 * b + 9], want[k]), (positive, b, k)
            # and the one-sample form on the UNPADDED question gives the same kept tokens (reference: no padding at all)
            short = lp.text_keep_batch(ids[b:b + 1, :n], types[b:b + 1, :n], cam_t[b, :n], is_positive_pert=positive)
            assert torch.equal(want[0][:, :n], short[0]) and torch.equal(want[2][:, :n], short[2]) and want[0][:, n:].abs().sum() == 0


def test_perturbation_driver_on_a_stand_in_model():
    """``LxmertPerturbation`` end to end on the host with a stand-in body whose score is an order-independent function of the
    regions it may attend to: 9 steps of B samples in one masked batch == the reference's loop over steps with the removed
    regions dropped; the zero-region step takes the region-free forward; the scoped GEMM selection is a no-op without a GPU."""
    import types

    from transformer_mm_explainability_amd import tuned_gemms

    class Body:
        def __call__(self, input_ids, visual_feats, visual_pos, attention_mask=None, token_type_ids=None, visual_attention_mask=None):
            m = torch.ones(visual_feats.shape[:2]) if visual_attention_mask is None else visual_attention_mask
            s = (visual_feats * m[..., None]).sum(1)[:, :5] + (visual_pos * m[..., None]).sum(1)[:, :1] + input_ids.float().sum(1, keepdim=True)
            return types.SimpleNamespace(question_answering_score=s)

    with tuned_gemms.scope("lxmert_pert") as on:
        assert on is False
    B, I = 3, 12
    g = torch.Generator().manual_seed(5)
    inputs = dict(input_ids=torch.randint(1, 50, (B, 7), generator=g), attention_mask=torch.ones(B, 7),
                  token_type_ids=torch.zeros(B, 7, dtype=torch.long), visual_feats=torch.randn(B, I, 8, generator=g),
                  visual_pos=torch.rand(B, I, 4, generator=g))
    cams = torch.rand(B, I, generator=g)
    body = Body()
    got = lp.LxmertPerturbation(body).perturbation_image(inputs, cams)
    assert got.shape == (B, 9, 5)
    for b in range(B):
        for s, step in enumerate(lp.PERT_STEPS):
            idx = cams[b].topk(k=int((1 - step) * I)).indices                      # perturbation.py:114-117
            want = body(inputs["input_ids"][b:b + 1], inputs["visual_feats"][b:b + 1, idx], inputs["visual_pos"][b:b + 1, idx])
            assert torch.allclose(got[b, s], want.question_answering_score[0], atol=1e-5)
    one = lp.LxmertPerturbation(body).perturbation_image({k: v[:1] for k, v in inputs.items()}, cams[0])
    assert one.shape == (9, 5) and torch.allclose(one, got[0], atol=1e-5)
