"""-m gpu: the MEASURED cfg-3 / cfg-4 legs (exactly what ``tools/bench_legs.py`` times) and the VisualBERT batch pass against
INDEPENDENT CPU oracle bodies (``oracle/detr_torch.py``, ``oracle/lxmert_torch.py``, ``oracle/visualbert_torch.py`` -- plain
torch autograd, each pinned on the reference's own model code in ``tests/test_oracle_golden.py``) at ``atol = 1e-5, rtol = 0``
(VERDICT r03 item 1).  Same seeds, sizes and wrappers as the bench legs: DETR-R50 head / 950 image tokens / K = 10 /
``rows_only`` / hipGraph; LXMERT-base / T = 14 / I = 36 / B = 32 / tape path / hipGraph.  Largest errors go to
``gpurun_out/parity_errors.json`` (-> ``profiles/r04_parity.json``).

Arg-max ties: the explained class / answer is ``argmax`` of logits computed in fp32 on two different machines; the tests take
the device's choice, check that it is an arg-max of the ORACLE's logits up to 1e-4, and explain that index on both sides."""
import numpy as np
import pytest
import torch

import parity

pytestmark = pytest.mark.gpu


def _cpu_sd(model):
    return {k: v.detach().cpu() for k, v in model.state_dict().items()}


def _is_argmax(logits_row, idx, tol=1e-4):
    return float(logits_row.max() - logits_row[idx]) <= tol


def test_cfg3_detr_r50_k10_rows_hipgraph_vs_oracle_body():
    """bench leg cfg 3: ``GraphedGenerateOursMulti(model, feats, K=10)`` (shared forward, hand-written batched backward,
    rows-only rules, hipGraph replay) == ``mask_generator.py:90-110``'s loop of ``Generator.generate_ours`` calls on the
    oracle body (matrix route: R_i_i 950 x 950, rules 6 / 7 / 10), row by row."""
    from oracle import detr_torch as dt
    from transformer_mm_explainability_amd import detr_model
    from transformer_mm_explainability_amd.detr_explainability import GraphedGenerateOursMulti
    torch.manual_seed(0)
    model = detr_model.detr_resnet50_head().cuda().eval()
    feats = torch.randn(1, 2048, 25, 38, device="cuda") * 0.5
    K = 10
    targets = torch.arange(K, device="cuda") * 3
    run = GraphedGenerateOursMulti(model, feats, K=K)
    run(feats, targets)
    out = run(feats, targets)                                         # a replay, like the timed loop
    torch.cuda.synchronize()
    with torch.no_grad():
        logits_dev = model(feats)["pred_logits"][0].cpu()
    classes = logits_dev[targets.cpu(), :-1].argmax(-1)

    sd = dt.prepare_state_dict(_cpu_sd(model))
    pos = dt.position_embedding_sine(torch.zeros(1, 25, 38, dtype=torch.bool), 128, normalize=True)
    logits, enc, dself, dcross = dt.forward(sd, feats.cpu(), pos, 8)
    parity.close(logits_dev, logits[0].detach(), atol=1e-4, what="pred_logits")
    ne, nd = len(enc), len(dself)
    a_enc, a_self, a_cross = dt._np(enc), dt._np(dself), dt._np(dcross)
    from oracle import relevancy_np as rn
    rows, rows64 = [], []
    for t, c in zip(targets.cpu().tolist(), classes.tolist()):
        assert _is_argmax(logits[0, t, :-1].detach(), c), "device arg-max is not an arg-max of the oracle's logits"
        grads = torch.autograd.grad(logits[0, t, c], enc + dself + dcross, retain_graph=True)
        rows.append(rn.detr_generate_ours_chain(a_enc, dt._np(grads[:ne]), a_self, dt._np(grads[ne:ne + nd]), a_cross,
                                                dt._np(grads[ne + nd:]), np.array([t]))[0, 0, 0])
        rows64.append(rn.detr_generate_ours_rows(a_enc, dt._np(grads[:ne]), a_self, dt._np(grads[ne:ne + nd]), a_cross,
                                                 dt._np(grads[ne + nd:]), np.array([t]))[0, 0, 0])
    want, ref64 = np.stack(rows), np.stack(rows64)
    assert out.shape == (1, 1, K, 950)
    # relative bound vs the fp32 ORACLE: measured 1.4e-4 of the largest entry (7.0e-8 on 5.1e-4, profiles/r04_parity.json).  Who owns
    # it: the fp64 evaluation of the same schedule on the ORACLE'S OWN slabs (``detr_generate_ours_rows``: float64
    # inside) is the referee -- the oracle's fp32 matrix route (R_i_i 950 x 950 through 6 layers, then diag(R_ii) - 1) sits ~1e-4 of
    # the largest entry away from it, the device rows ~2e-5 (both recorded below: profiles/rNN_parity.json).  So the oracle side owns
    # the error; the device result is held to the suite's 1e-4 against the REFEREE and to 3e-4 against the fp32 oracle.
    parity.close(out[0, 0], want, atol=1e-5, rtol=0.0, what="R_q_i rows (K=10, Ni=950)", relmax=3e-4)
    parity.close(out[0, 0], ref64, atol=1e-5, rtol=0.0, what="R_q_i rows (K=10, Ni=950) vs the fp64 referee", relmax=1e-4)
    top = float(np.abs(ref64).max())
    parity.note("fp32 oracle (matrix route) vs the fp64 referee, same slabs", float(np.abs(want - ref64).max()), None, top)


def test_cfg4_lxmert_base_b32_tape_hipgraph_vs_oracle_body():
    """bench leg cfg 4 (explain half): ``GraphedGenerateOursBatch(model, batch)`` -- LXMERT-base, B = 32, T = 14, I = 36, tape
    forward / backward, one schedule launch, hipGraph replay -- == ``GeneratorOurs.generate_ours`` per item on the oracle body."""
    from oracle import lxmert_torch as lt
    from transformer_mm_explainability_amd import lxmert_explainability as le
    from transformer_mm_explainability_amd import lxmert_model as lm
    torch.manual_seed(0)
    model = lm.LxmertForQuestionAnswering(lm.LxmertConfig()).cuda().eval()
    B, T, I = 32, 14, 36
    gb = torch.Generator().manual_seed(2)
    cpu = dict(input_ids=torch.randint(1, 30000, (B, T), generator=gb), attention_mask=torch.ones(B, T),
               token_type_ids=torch.zeros(B, T, dtype=torch.long),
               visual_feats=torch.randn(B, I, 2048, generator=gb), visual_pos=torch.rand(B, I, 4, generator=gb))
    batch = {k: v.cuda() for k, v in cpu.items()}
    run = le.GraphedGenerateOursBatch(model, batch)
    run(batch)
    R_t_t, R_t_i = run(batch)
    torch.cuda.synchronize()
    with torch.no_grad():
        score_dev = model(**batch).question_answering_score.cpu()
    answers = score_dev.argmax(-1)

    sd = lt.prepare_state_dict(_cpu_sd(model))
    want_tt, want_ti = [], []
    for b in range(B):
        item = {k: v[b:b + 1] for k, v in cpu.items()}
        score, _ = lt.forward(sd, 12, **item)
        assert _is_argmax(score[0].detach(), int(answers[b])), "device arg-max is not an arg-max of the oracle's scores"
        parity.close(score_dev[b], score[0].detach(), atol=1e-4, what="answer scores")
        tt, ti = lt.generate_ours(sd, 12, item, index=int(answers[b]))
        want_tt.append(tt)
        want_ti.append(ti)
    parity.close(R_t_t, np.stack(want_tt), atol=1e-5, rtol=0.0, what="R_t_t (B=32, T=14)")
    parity.close(R_t_i, np.stack(want_ti), atol=1e-5, rtol=0.0, what="R_t_i (B=32, T=14, I=36)")


def test_visualbert_base_b8_tape_hipgraph_vs_oracle_body():
    """``visualbert_explainability.GraphedGenerateOursBatch`` -- BERT-base stack, 12 text tokens + 100 regions, B = 8, tape path,
    one chain launch, hipGraph replay -- == ``SelfAttentionGenerator.generate_ours`` per item on the oracle body."""
    from oracle import visualbert_torch as vt
    from transformer_mm_explainability_amd import visualbert_explainability as ve
    from transformer_mm_explainability_amd import visualbert_model as vm
    torch.manual_seed(0)
    model = vm.VisualBERT(vm.VisualBertConfig()).cuda().eval()
    B, Tpad, T, V = 8, 16, 12, 100
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(1, 30000, (B, Tpad), generator=g)
    ids[:, T:] = 0
    mask = torch.zeros(B, Tpad, dtype=torch.long)
    mask[:, :T] = 1
    cpu = dict(input_ids=ids, input_mask=mask, segment_ids=torch.zeros(B, Tpad, dtype=torch.long),
               image_feature_0=torch.randn(B, V, 2048, generator=g))
    sample = {k: v.cuda() for k, v in cpu.items()}
    run = ve.GraphedGenerateOursBatch(model, sample)
    run(sample)
    out = run(sample)
    torch.cuda.synchronize()
    with torch.no_grad():
        scores_dev = torch.cat([model({k: v[b:b + 1].clone() for k, v in sample.items()})["scores"] for b in range(B)]).cpu()
    answers = scores_dev.argmax(-1)

    sd = vt.prepare_state_dict(_cpu_sd(model))
    want = []
    for b in range(B):
        scores, _ = vt.forward(sd, 12, ids[b:b + 1], mask[b:b + 1], cpu["image_feature_0"][b:b + 1])
        assert _is_argmax(scores[0].detach(), int(answers[b]))
        parity.close(scores_dev[b], scores[0].detach(), atol=1e-4, what="scores")
        want.append(vt.generate_ours(sd, 12, ids[b:b + 1], mask[b:b + 1], cpu["image_feature_0"][b:b + 1],
                                     index=int(answers[b]))[0])
    parity.close(out, np.stack(want), atol=1e-5, rtol=0.0, what="cls row (B=8, N=112)")
