"""-m gpu: the batched perturbation evaluator (section 8f row 2) == the reference's sequential loop
(lxmert/lxmert/perturbation.py:110-194, restated here: physically remove regions / tokens, one forward per step)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model_and_inputs(T=12, I=20):
    from transformer_mm_explainability_amd import lxmert_model as lm
    torch.manual_seed(5)
    cfg = lm.LxmertConfig(hidden_size=96, num_attention_heads=4, intermediate_size=192, l_layers=3, x_layers=2,
                          r_layers=2, visual_feat_dim=40, vocab_size=200, num_qa_labels=31,
                          max_position_embeddings=64)
    model = lm.LxmertForQuestionAnswering(cfg).cuda().eval()
    with torch.no_grad():                                     # default init leaves the scores nearly flat
        for p in model.parameters():
            if p.dim() > 1:
                p.mul_(3.0)
    g = torch.Generator().manual_seed(6)
    inputs = dict(input_ids=torch.randint(1, 200, (1, T), generator=g).cuda(),
                  attention_mask=torch.ones(1, T).cuda(), token_type_ids=torch.zeros(1, T, dtype=torch.long).cuda(),
                  visual_feats=torch.randn(1, I, 40, generator=g).cuda(), visual_pos=torch.rand(1, I, 4, generator=g).cuda())
    return model, inputs, g


@pytest.mark.parametrize("positive", [False, True])
def test_image_perturbation_batched_equals_sequential(positive):
    from transformer_mm_explainability_amd import lxmert_perturbation as lp
    model, inputs, g = _model_and_inputs()
    cam = torch.rand(20, generator=g).cuda()
    got = lp.LxmertPerturbation(model).perturbation_image(inputs, cam, positive)
    c = -cam if positive else cam
    with torch.no_grad():
        for s, step in enumerate(lp.PERT_STEPS):
            idx = c.topk(k=int((1 - step) * 20), dim=-1).indices
            want = model(input_ids=inputs["input_ids"], attention_mask=inputs["attention_mask"],
                         token_type_ids=inputs["token_type_ids"], visual_feats=inputs["visual_feats"][:, idx],
                         visual_pos=inputs["visual_pos"][:, idx]).question_answering_score[0]
            torch.testing.assert_close(got[s], want, rtol=1e-4, atol=1e-5)
            assert got[s].argmax() == want.argmax()
    assert not torch.allclose(got[0], got[-1], atol=1e-3)     # the perturbation does change the answer scores


@pytest.mark.parametrize("positive", [False, True])
def test_text_perturbation_batched_equals_sequential(positive):
    from transformer_mm_explainability_amd import lxmert_perturbation as lp
    model, inputs, g = _model_and_inputs()
    T = inputs["input_ids"].shape[1]
    cam = torch.rand(T, generator=g).cuda()
    got = lp.LxmertPerturbation(model).perturbation_text(inputs, cam, positive)
    c = -cam if positive else cam
    with torch.no_grad():
        for s, step in enumerate(lp.PERT_STEPS):
            pure = c[1:-1]
            top = pure.topk(k=int((1 - step) * (T - 2)), dim=-1).indices.tolist()
            kept = sorted([0, T - 1] + [i + 1 for i in top])
            want = model(input_ids=inputs["input_ids"][:, kept], attention_mask=inputs["attention_mask"][:, kept],
                         token_type_ids=inputs["token_type_ids"][:, kept], visual_feats=inputs["visual_feats"],
                         visual_pos=inputs["visual_pos"]).question_answering_score[0]
            torch.testing.assert_close(got[s], want, rtol=1e-4, atol=1e-5)
    labels = torch.zeros(31, device="cuda")
    labels[got[0].argmax()] = 1.0
    acc = lp.LxmertPerturbation.accuracy(got, labels)
    assert acc.shape == (9,) and acc[0] == 1.0


def test_generator_to_perturbation_pipeline():
    """relevancy -> normalised cams -> both perturbation tests, all on device (perturbation.py:216-250 for one item)."""
    import types
    from transformer_mm_explainability_amd import lxmert_explainability as le
    from transformer_mm_explainability_amd import lxmert_perturbation as lp
    model, inputs, _ = _model_and_inputs()
    usage = types.SimpleNamespace(model=model, text_len=12, image_boxes_len=20, forward=lambda item: model(**inputs))
    R_t_t, R_t_i = le.GeneratorOurs(usage).generate_ours(None, use_lrp=False)
    cam_image, cam_text = lp.normalize_cams(R_t_t, R_t_i)
    pert = lp.LxmertPerturbation(model)
    for scores in (pert.perturbation_image(inputs, cam_image), pert.perturbation_text(inputs, cam_text)):
        assert scores.shape == (9, 31) and torch.isfinite(scores).all()


def test_generate_ours_batch_equals_per_item_loop():
    """B items of equal question length in one forward/backward/schedule launch == the evaluator's per-item calls."""
    import types
    from transformer_mm_explainability_amd import lxmert_explainability as le
    model, inputs, g = _model_and_inputs()
    B = 5
    batch = dict(input_ids=torch.randint(1, 200, (B, 12), generator=g).cuda(), attention_mask=torch.ones(B, 12).cuda(),
                 token_type_ids=torch.zeros(B, 12, dtype=torch.long).cuda(),
                 visual_feats=torch.randn(B, 20, 40, generator=g).cuda(), visual_pos=torch.rand(B, 20, 4, generator=g).cuda())
    gen = le.GeneratorOurs(types.SimpleNamespace(model=model, text_len=12, image_boxes_len=20))
    R_tt, R_ti = gen.generate_ours_batch(batch)
    assert R_tt.shape == (B, 12, 12) and R_ti.shape == (B, 12, 20)
    for b in range(B):
        one = {k: v[b:b + 1] for k, v in batch.items()}
        usage = types.SimpleNamespace(model=model, text_len=12, image_boxes_len=20, forward=lambda item: model(**one))
        tt, ti = le.GeneratorOurs(usage).generate_ours(None, use_lrp=False)
        torch.testing.assert_close(R_tt[b], tt, rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(R_ti[b], ti, rtol=1e-4, atol=1e-6)


def test_perturbation_batched_over_items():
    from transformer_mm_explainability_amd import lxmert_perturbation as lp
    model, inputs, g = _model_and_inputs()
    B = 3
    batch = dict(input_ids=torch.randint(1, 200, (B, 12), generator=g).cuda(), attention_mask=torch.ones(B, 12).cuda(),
                 token_type_ids=torch.zeros(B, 12, dtype=torch.long).cuda(),
                 visual_feats=torch.randn(B, 20, 40, generator=g).cuda(), visual_pos=torch.rand(B, 20, 4, generator=g).cuda())
    cam_i, cam_t = torch.rand(B, 20, generator=g).cuda(), torch.rand(B, 12, generator=g).cuda()
    pert = lp.LxmertPerturbation(model)
    img, txt = pert.perturbation_image(batch, cam_i), pert.perturbation_text(batch, cam_t, True)
    assert img.shape == txt.shape == (B, 9, 31)
    for b in range(B):
        one = {k: v[b:b + 1] for k, v in batch.items()}
        torch.testing.assert_close(img[b], pert.perturbation_image(one, cam_i[b]), rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(txt[b], pert.perturbation_text(one, cam_t[b], True), rtol=1e-4, atol=1e-5)
    labels = torch.rand(B, 31, device="cuda")
    acc = lp.LxmertPerturbation.accuracy(img, labels)
    assert acc.shape == (B, 9) and acc[1, 0] == labels[1, img[1, 0].argmax()]


def _visualbert_and_sample():
    from transformer_mm_explainability_amd import visualbert_model as vm
    torch.manual_seed(9)
    cfg = vm.VisualBertConfig(hidden_size=96, num_attention_heads=4, intermediate_size=192, num_hidden_layers=3,
                              vocab_size=300, max_position_embeddings=64, visual_embedding_dim=40, num_labels=23)
    model = vm.VisualBERT(cfg).cuda().eval()
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() > 1:
                p.mul_(3.0)
    g = torch.Generator().manual_seed(10)
    T, n_text, V = 16, 12, 14
    ids = torch.randint(1, 300, (1, T), generator=g)
    ids[0, n_text:] = 0
    mask = torch.zeros(1, T, dtype=torch.long)
    mask[0, :n_text] = 1

    def sample():
        return {"input_ids": ids.cuda(), "input_mask": mask.cuda(), "segment_ids": torch.zeros(1, T, dtype=torch.long).cuda(),
                "image_feature_0": feats}

    feats = torch.randn(1, V, 40, generator=g).cuda()
    return model, sample, n_text, V, g


@pytest.mark.parametrize("positive", [False, True])
def test_visualbert_perturbation_equals_sequential(positive):
    """evaluation_loop.py:100-166 restated (physical gathers, one forward per step through the VisualBERT wrapper)
    == the batched evaluator, for the generator's own relevancy row."""
    from transformer_mm_explainability_amd import visualbert_explainability as vb
    from transformer_mm_explainability_amd import visualbert_perturbation as vp
    model, sample, n_text, V, g = _visualbert_and_sample()
    cam = vb.SelfAttentionGenerator(model).generate_ours(sample()).detach()          # [1, n_text + V]
    assert cam.shape == (1, n_text + V)
    pert = vp.VisualBertPerturbation(model)
    got_img = pert.perturbation_image(sample(), cam, positive)
    got_txt = pert.perturbation_text(sample(), cam, positive)
    c = -cam if positive else cam
    cls_index = n_text - 2
    with torch.no_grad():
        for s, (step, step_txt) in enumerate(zip(pert.image_steps, pert.text_steps)):
            sl = sample()
            idx = c[0, n_text:].topk(k=int((1 - step) * V), dim=-1).indices
            sl["image_feature_0"] = sl["image_feature_0"][:, idx]
            if idx.numel():
                torch.testing.assert_close(got_img[s], model(sl)["scores"][0], rtol=1e-4, atol=1e-5)
            sl = sample()
            scores = c[0, 1:cls_index]
            top = scores.topk(k=int((1 - step_txt) * scores.shape[0]), dim=-1).indices.tolist()
            kept = sorted([0, cls_index, cls_index + 1] + [i + 1 for i in top])
            T = sl["input_ids"].shape[1]
            sl["input_ids"] = torch.cat((sl["input_ids"][:, kept], sl["input_ids"][:, n_text:]), dim=1)
            sl["input_mask"] = torch.cat((sl["input_mask"][:, kept], sl["input_mask"][:, n_text:]), dim=1)
            sl["segment_ids"] = sl["segment_ids"][:, :sl["input_ids"].shape[1]]
            torch.testing.assert_close(got_txt[s], model(sl)["scores"][0], rtol=1e-4, atol=1e-5)
    assert torch.isfinite(got_img).all() and not torch.allclose(got_img[0], got_img[-1], atol=1e-3)


def test_visualbert_generate_ours_batch_equals_per_item():
    from transformer_mm_explainability_amd import visualbert_explainability as vb
    model, sample, n_text, V, g = _visualbert_and_sample()
    B, T = 4, 16
    ids = torch.randint(1, 300, (B, T), generator=g)
    ids[:, n_text:] = 0
    mask = torch.zeros(B, T, dtype=torch.long)
    mask[:, :n_text] = 1
    feats = torch.randn(B, V, 40, generator=g)
    batch = {"input_ids": ids.cuda(), "input_mask": mask.cuda(), "segment_ids": torch.zeros(B, T, dtype=torch.long).cuda(),
             "image_feature_0": feats.cuda()}
    got = vb.SelfAttentionGenerator(model).generate_ours_batch(batch)
    assert got.shape == (B, n_text + V)
    for b in range(B):
        one = {k: v[b:b + 1].clone() for k, v in batch.items()}
        want = vb.SelfAttentionGenerator(model).generate_ours(one)
        torch.testing.assert_close(got[b:b + 1], want, rtol=1e-4, atol=1e-6)


# ----------------------------------------------------------------------------------------------------------------------
# Against the REFERENCE's own evaluator code: tests/golden/{lxmert,visualbert}_perturbation.npz hold
# the answer scores of every one of the 9 re-runs that ``ModelPert.perturbation_image/_text`` (lxmert/lxmert/perturbation.py:
# 85-194) and ``TrainerEvaluationLoopMixinPert.evaluation_loop`` (VisualBERT/.../evaluation_loop.py:73-169) -- exec'd from
# the reference files -- produced on the reference bodies (weights: lxmert_model.npz / visualbert_model.npz).
# ----------------------------------------------------------------------------------------------------------------------
def _cu(x):
    import numpy as np
    return torch.from_numpy(np.asarray(x)).cuda()


@pytest.mark.parametrize("modality", ["image", "text"])
@pytest.mark.parametrize("positive", [False, True])
def test_lxmert_perturbation_vs_reference_evaluator(golden, modality, positive):
    from transformer_mm_explainability_amd import lxmert_explainability as le
    from transformer_mm_explainability_amd import lxmert_perturbation as lp
    from transformer_mm_explainability_amd import lxmert_model as lm
    import types
    gm, g = golden("lxmert_model"), golden("lxmert_perturbation")
    hidden, heads, inter, ll, xl, rl, feat, vocab, labels, max_pos, T, I = (int(x) for x in gm["dims"])
    model = lm.LxmertForQuestionAnswering(lm.LxmertConfig(
        hidden_size=hidden, num_attention_heads=heads, intermediate_size=inter, l_layers=ll, x_layers=xl, r_layers=rl,
        visual_feat_dim=feat, vocab_size=vocab, num_qa_labels=labels, max_position_embeddings=max_pos))
    model.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in gm.items() if k.startswith("w__")}, strict=True)
    model = model.cuda().eval()
    inputs = {k[4:]: _cu(v) for k, v in gm.items() if k.startswith("in__")}
    usage = types.SimpleNamespace(model=model, text_len=T, image_boxes_len=I, forward=lambda item: model(**inputs))
    # the cams the reference evaluator used are the normalised first rows of ITS generator's maps; ours must agree
    R_t_t, R_t_i = le.GeneratorOurs(usage).generate_ours(None, use_lrp=False)
    cam_image, cam_text = lp.normalize_cams(R_t_t, R_t_i)
    torch.testing.assert_close(cam_image.cpu(), torch.from_numpy(g["cam_image"]), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(cam_text.cpu(), torch.from_numpy(g["cam_text"]), rtol=1e-4, atol=1e-5)
    pert = lp.LxmertPerturbation(model)
    tag = "%s_%s" % (modality, "pos" if positive else "neg")
    if modality == "image":
        scores = pert.perturbation_image(inputs, _cu(g["cam_image"]), positive)
    else:
        scores = pert.perturbation_text(inputs, _cu(g["cam_text"]), positive)
    torch.testing.assert_close(scores.cpu(), torch.from_numpy(g["scores_" + tag]), rtol=1e-4, atol=1e-5)
    acc = pert.accuracy(scores, _cu(g["label_scores"]))
    torch.testing.assert_close(acc.cpu(), torch.from_numpy(g["acc_" + tag]), rtol=0, atol=1e-6)


def _visualbert_from_golden(g):
    from transformer_mm_explainability_amd import visualbert_model as vm
    hidden, heads, inter, layers, vocab, max_pos, vdim, labels = (int(x) for x in g["dims"])
    model = vm.VisualBERT(vm.VisualBertConfig(hidden_size=hidden, num_attention_heads=heads, intermediate_size=inter,
                                              num_hidden_layers=layers, vocab_size=vocab, max_position_embeddings=max_pos,
                                              visual_embedding_dim=vdim, num_labels=labels))
    model.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("w__")}, strict=False)
    return model.cuda().eval()


@pytest.mark.parametrize("modality", ["image", "text"])
@pytest.mark.parametrize("positive", [False, True])
def test_visualbert_perturbation_vs_reference_evaluator(golden, modality, positive):
    """Scores of all 3 x 9 re-runs and the numbers the reference loop PRINTS (its ``i > num_samples`` accounting: three
    items evaluated, divided by ``num_samples = 2``) -- ``reference_exact=True``; the default counts ``num_samples`` items."""
    from transformer_mm_explainability_amd import visualbert_explainability as vb
    from transformer_mm_explainability_amd import visualbert_perturbation as vp
    gm, g = golden("visualbert_model"), golden("visualbert_perturbation")
    model = _visualbert_from_golden(gm)
    feats = [_cu(gm["image_feature_0"])] + [_cu(f[None]) for f in g["image_features_1_2"]]

    def item(i):
        return {"input_ids": _cu(gm["input_ids"]), "input_mask": _cu(gm["input_mask"]),
                "segment_ids": torch.zeros_like(_cu(gm["input_ids"])), "image_feature_0": feats[i],
                "targets": _cu(g["targets"])}

    tag = "%s_%s" % (modality, "pos" if positive else "neg")
    gen = vb.SelfAttentionGenerator(model)
    pert = vp.VisualBertPerturbation(model)
    for i in range(3):
        cam = gen.generate_ours(item(i)).detach()
        run = pert.perturbation_image if modality == "image" else pert.perturbation_text
        scores = run(item(i), cam, positive)
        torch.testing.assert_close(scores.cpu(), torch.from_numpy(g["scores_" + tag][i]), rtol=1e-4, atol=1e-5)
    printed = vp.evaluation_loop(gen.generate_ours, pert, [item(0), item(1), item(2)], num_samples=2, modality=modality,
                                 is_positive_pert=positive, reference_exact=True)
    torch.testing.assert_close(printed, torch.from_numpy(g["printed_step_acc_" + tag]), rtol=0, atol=1e-4)
    plain = vp.evaluation_loop(gen.generate_ours, pert, [item(0), item(1), item(2)], num_samples=2, modality=modality,
                               is_positive_pert=positive)
    assert (plain <= printed + 1e-9).all()                    # two items instead of three, same divisor


def test_ranking_tie_policy_is_stable_and_prefix_consistent():
    """with tied scores ``topk(k)`` for different k need not be nested; the evaluator ranks ONCE with a stable
    sort (ties -> lower index first), so every step keeps a prefix of the same order, and for distinct scores the kept
    sets equal the reference's per-step ``topk``."""
    from transformer_mm_explainability_amd import lxmert_perturbation as lp
    cam = torch.tensor([0.5, 0.0, 0.5, 1.0, 0.0, 0.5, 0.0, 0.25], device="cuda")
    keep = lp.image_keep_masks(cam)
    assert lp.ranking(cam).tolist() == [3, 0, 2, 5, 7, 1, 4, 6]
    for s in range(1, keep.shape[0]):
        assert (keep[s] <= keep[s - 1]).all()                  # nested kept sets
    g = torch.Generator().manual_seed(0)
    cam = torch.rand(36, generator=g).cuda()
    keep = lp.image_keep_masks(cam)
    for s, step in enumerate(lp.PERT_STEPS):
        want = torch.zeros(36, device="cuda")
        want[cam.topk(int((1 - step) * 36)).indices] = 1
        assert torch.equal(keep[s], want)


# ----------------------------------------------------------------------------------------------------------------------
# Ragged question lengths inside ONE padded batch + the hipGraph replay of the batched explain pass
# ----------------------------------------------------------------------------------------------------------------------
def _ragged_batch(g, lens, I=20, T=12):
    B = len(lens)
    ids = torch.zeros(B, T, dtype=torch.long)
    mask = torch.zeros(B, T)
    for b, n in enumerate(lens):
        ids[b, :n] = torch.randint(1, 200, (n,), generator=g)
        mask[b, :n] = 1
    return dict(input_ids=ids.cuda(), attention_mask=mask.cuda(), token_type_ids=torch.zeros(B, T, dtype=torch.long).cuda(),
                visual_feats=torch.randn(B, I, 40, generator=g).cuda(), visual_pos=torch.rand(B, I, 4, generator=g).cuda())


def test_generate_ours_batch_with_per_sample_question_lengths():
    """A batch padded to T = 12 with questions of 5 / 12 / 8 / 9 tokens == each item explained alone, unpadded (the way
    the reference's evaluator calls the generator); rows / columns beyond a sample's length come back as zero."""
    import types
    from transformer_mm_explainability_amd import lxmert_explainability as le
    model, _, g = _model_and_inputs()
    lens = [5, 12, 8, 9]
    batch = _ragged_batch(g, lens)
    gen = le.GeneratorOurs(types.SimpleNamespace(model=model))
    R_t_t, R_t_i = gen.generate_ours_batch(batch)
    assert R_t_t.shape == (4, 12, 12) and R_t_i.shape == (4, 12, 20)
    for b, n in enumerate(lens):
        one = {k: v[b:b + 1, :n] if k in ("input_ids", "attention_mask", "token_type_ids") else v[b:b + 1]
               for k, v in batch.items()}
        usage = types.SimpleNamespace(model=model, text_len=n, image_boxes_len=20, forward=lambda item: model(**one))
        want_tt, want_ti = le.GeneratorOurs(usage).generate_ours(None, use_lrp=False)
        torch.testing.assert_close(R_t_t[b, :n, :n], want_tt, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(R_t_i[b, :n], want_ti, rtol=1e-4, atol=1e-5)
        assert (R_t_t[b, n:] == 0).all() and (R_t_t[b, :, n:] == 0).all() and (R_t_i[b, n:] == 0).all()


def test_graphed_generate_ours_batch_replays_any_lengths():
    """ONE captured graph (padded length 12) serves batches of different question lengths; the deferred diag word is
    checked once per batch and nothing else synchronises."""
    import types
    from transformer_mm_explainability_amd import lxmert_explainability as le
    from transformer_mm_explainability_amd import lxmert_perturbation as lp
    model, _, g = _model_and_inputs()
    first = _ragged_batch(g, [12, 12, 12, 12])
    run = le.GraphedGenerateOursBatch(model, first)
    gen = le.GeneratorOurs(types.SimpleNamespace(model=model))
    for lens in ([12, 12, 12, 12], [6, 11, 9, 7]):
        batch = _ragged_batch(g, lens)
        want_tt, want_ti = (t.clone() for t in gen.generate_ours_batch(batch))
        got_tt, got_ti = run(batch)
        torch.testing.assert_close(got_tt, want_tt, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(got_ti, want_ti, rtol=1e-5, atol=1e-6)
        assert float(run.diag_min) >= 0
        # the evaluator's next step on the padded batch: cams over real tokens only, both perturbation tests
        cam_image, cam_text = lp.normalize_cams_batch(got_tt, got_ti, batch["attention_mask"])
        assert torch.isfinite(cam_image).all() and torch.isfinite(cam_text).all()
        pert = lp.LxmertPerturbation(model)
        img = pert.perturbation_image(batch, cam_image)
        txt = pert.perturbation_text(batch, cam_text)
        assert img.shape == txt.shape == (4, 9, 31)
        b, n = 1, lens[1]                       # the padded text test == the unpadded one of the same item
        one = {k: v[b:b + 1, :n] if k in ("input_ids", "attention_mask", "token_type_ids") else v[b:b + 1]
               for k, v in batch.items()}
        torch.testing.assert_close(txt[b], pert.perturbation_text(one, cam_text[b, :n]), rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(img[b], pert.perturbation_image(one, cam_image[b]), rtol=1e-4, atol=1e-5)
    with pytest.raises(ValueError, match="captured"):
        run(_ragged_batch(g, [5, 5, 5]))
    # round 6: the 9-step image test of a fixed-shape batch as ONE hipGraph (GraphedImagePerturbation) == the eager calls
    labels = torch.rand(4, 31, device="cuda")
    graphed = None
    for lens in ([12, 12, 12, 12], [6, 11, 9, 7], [12, 5, 12, 8]):
        batch = _ragged_batch(g, lens)
        tt, ti = (t.clone() for t in run(batch))
        cam_image, _ = lp.normalize_cams_batch(tt, ti, batch["attention_mask"])
        want = lp.LxmertPerturbation.accuracy(lp.LxmertPerturbation(model, tuned=False).perturbation_image(batch, cam_image), labels)
        if graphed is None:
            graphed = lp.GraphedImagePerturbation(lp.LxmertPerturbation(model), batch, tt, ti, labels)
        got = graphed(batch, tt, ti, labels)
        assert got.shape == (4, 9) and torch.equal(got, want), (got, want)


def test_lxmert_tape_path_equals_autograd_path():
    """``LxmertForQuestionAnswering.forward_tape`` / ``backward_tape`` (hand-written vector-Jacobian chain, packed q/k/v GEMMs,
    fused add + LayerNorm, ``bert_tape.py``) vs the autograd route through the same body: answer scores, every attention
    block's captured probabilities and gradient slabs, and the relevancies, on a ragged batch."""
    import types
    from transformer_mm_explainability_amd import lxmert_explainability as le
    model, _, g = _model_and_inputs()
    batch = _ragged_batch(g, [5, 12, 8, 9])
    want_scores = model(**batch).question_answering_score
    got_scores, _ = model.forward_tape(**batch)
    torch.testing.assert_close(got_scores, want_scores, rtol=1e-5, atol=1e-5)
    gen_a, gen_t = le.GeneratorOurs(types.SimpleNamespace(model=model)), le.GeneratorOurs(types.SimpleNamespace(model=model))
    gen_a.use_tape = False
    want = [t.clone() for t in gen_a.generate_ours_batch(batch)]
    enc = model.lxmert.encoder
    mods = [b.attention.self for b in list(enc.layer) + list(enc.r_layers)]
    for x in enc.x_layers:
        mods += [x.visual_attention.att, x.lang_self_att.self]
    for x in list(enc.x_layers)[:-1]:
        mods += [x.visual_attention_copy.att, x.visn_self_att.self]
    slabs = [(m.get_attn().clone(), m.get_attn_gradients().clone()) for m in mods]
    got = gen_t.generate_ours_batch(batch)
    for m, (p, dp) in zip(mods, slabs):
        torch.testing.assert_close(m.get_attn(), p, rtol=1e-5, atol=2e-6)
        scale = float(dp.abs().max())
        assert float((m.get_attn_gradients() - dp).abs().max()) <= 1e-5 * scale + 1e-9
    for a, b in zip(got, want):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6)


def test_lxmert_two_stream_tape_path_equals_one_stream():
    """``LxmertEncoder.overlap_modalities`` (the image chain of every layer group on a side stream beside the text chain) is a
    schedule: the forward gives the same bits, slabs and relevancies equal the one-stream tape path to fp32 rounding; the
    perturbation re-runs through ``scores_no_grad`` equal the module forward (to fp32 rounding: other GEMM shapes)."""
    import types
    from transformer_mm_explainability_amd import lxmert_explainability as le
    from transformer_mm_explainability_amd import lxmert_perturbation as lp
    model, _, g = _model_and_inputs()
    batch = _ragged_batch(g, [5, 12, 8, 9])
    enc = model.lxmert.encoder
    outs = []
    try:
        for overlap in (True, False):
            enc.overlap_modalities = overlap
            gen = le.GeneratorOurs(types.SimpleNamespace(model=model))
            R = [t.clone() for t in gen.generate_ours_batch(batch)]
            slabs = [b.attention.self.get_attn_gradients().clone() for b in list(enc.layer) + list(enc.r_layers)]
            slabs += [x.visual_attention_copy.att.get_attn_gradients().clone() for x in list(enc.x_layers)[:-1]]
            outs.append((R, slabs, model.scores_no_grad(**batch).clone()))
            torch.cuda.synchronize()
    finally:
        enc.overlap_modalities = True
    assert torch.equal(outs[0][2], outs[1][2])                        # the forward: the same kernels in another order
    for a, b in zip(outs[0][0], outs[1][0]):                          # (the backward joins the two chains with an add where the
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6)        # one-stream path accumulates inside a GEMM: rounding)
    for a, b in zip(outs[0][1], outs[1][1]):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-9
    torch.testing.assert_close(outs[0][2], model(**batch).question_answering_score, rtol=1e-5, atol=1e-5)
    cams = torch.rand(4, batch["visual_feats"].shape[1], generator=g).cuda()
    fast = lp.LxmertPerturbation(model).perturbation_image(batch, cams)
    class Plain:                                                     # a body without the fast forward: the module route
        def __call__(self, **kw):
            return model(**kw)
    torch.testing.assert_close(fast, lp.LxmertPerturbation(Plain()).perturbation_image(batch, cams), rtol=1e-4, atol=1e-5)


def test_visualbert_graphed_generate_ours_batch():
    """hipGraph replay of the batched VisualBERT explain pass (tape path) == the eager call, also after new inputs were
    copied in; a batch of another text length is refused."""
    from transformer_mm_explainability_amd import visualbert_explainability as vb
    model, sample, n_text, V, g = _visualbert_and_sample()
    B, T = 3, 16

    def make():
        ids = torch.randint(1, 300, (B, T), generator=g)
        ids[:, n_text:] = 0
        mask = torch.zeros(B, T, dtype=torch.long)
        mask[:, :n_text] = 1
        return {"input_ids": ids.cuda(), "input_mask": mask.cuda(), "segment_ids": torch.zeros(B, T, dtype=torch.long).cuda(),
                "image_feature_0": torch.randn(B, V, 40, generator=g).cuda()}
    first, second = make(), make()
    run = vb.GraphedGenerateOursBatch(model, first)
    gen = vb.SelfAttentionGenerator(model)
    for batch in (first, second):
        want = gen.generate_ours_batch({k: v.clone() for k, v in batch.items()}).clone()
        got = run(batch)
        torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-7)
    gen.use_tape = False                                                   # the autograd route of the same body
    torch.testing.assert_close(run(second), gen.generate_ours_batch({k: v.clone() for k, v in second.items()}), rtol=1e-4, atol=1e-6)
    short = make()
    short["input_mask"][:, n_text - 1] = 0
    with pytest.raises(ValueError, match="text tokens"):
        run(short)
