"""cfg-4 image perturbation test (LXMERT-base, B = 32, T = 14 / 20, 36 regions): all live steps as ONE masked batch (rounds 1-5) vs the two
gathered step groups (round 6), eager with the tuned GEMM selection and replayed from a hipGraph (what the evaluator runs); interleaved."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformer_mm_explainability_amd import lxmert_model as lm  # noqa: E402
from transformer_mm_explainability_amd import lxmert_perturbation as lp  # noqa: E402

torch.manual_seed(0)
model = lm.LxmertForQuestionAnswering(lm.LxmertConfig()).cuda().eval()
B, I = 32, 36


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for T in (14, 20):
    g = torch.Generator().manual_seed(2)
    batch = dict(input_ids=torch.randint(1, 30000, (B, T), generator=g).cuda(), attention_mask=torch.ones(B, T).cuda(),
                 token_type_ids=torch.zeros(B, T, dtype=torch.long).cuda(),
                 visual_feats=torch.randn(B, I, 2048, generator=g).cuda(), visual_pos=torch.rand(B, I, 4, generator=g).cuda())
    R_t_t, R_t_i = torch.rand(B, T, T, generator=g).cuda(), torch.rand(B, T, I, generator=g).cuda()
    labels = torch.rand(B, model.config.num_qa_labels, generator=g).cuda()
    cams = torch.rand(B, I, generator=g).cuda()
    perts = {grouped: lp.LxmertPerturbation(model, grouped=grouped) for grouped in (False, True)}
    a, b = perts[False].perturbation_image(batch, cams), perts[True].perturbation_image(batch, cams)
    print("T=%d  max |scores(one masked batch) - scores(two gathered groups)| = %.3e (max |score| %.3f); arg-max equal: %s"
          % (T, float((a - b).abs().max()), float(a.abs().max()), bool((a.argmax(-1) == b.argmax(-1)).all())))
    graphs = {grouped: lp.GraphedImagePerturbation(perts[grouped], batch, R_t_t, R_t_i, labels) for grouped in (False, True)}
    res = {}
    for rnd in range(3):
        for grouped in (False, True):
            res.setdefault(("eager tuned", grouped), []).append(timed(lambda: perts[grouped].perturbation_image(batch, cams), 5))
            res.setdefault(("hipGraph", grouped), []).append(timed(lambda: graphs[grouped](batch, R_t_t, R_t_i, labels), 10))
    for (how, grouped), v in sorted(res.items()):
        print("T=%d  %-12s %-22s median %6.2f ms (min %.2f max %.2f)" % (T, how, "two gathered groups" if grouped else "one masked batch",
                                                                        statistics.median(v), min(v), max(v)))
