"""LXMERT-base sized perturbation evaluator (config 4): 9 sequential forwards (reference loop) vs the batched form."""
import sys
import time

import torch

sys.path.insert(0, ".")
from transformer_mm_explainability_amd import lxmert_model as lm  # noqa: E402
from transformer_mm_explainability_amd import lxmert_perturbation as lp  # noqa: E402

torch.manual_seed(0)
model = lm.LxmertForQuestionAnswering(lm.LxmertConfig()).cuda().eval()
T, I = 14, 36
g = torch.Generator().manual_seed(1)
inputs = dict(input_ids=torch.randint(1, 30000, (1, T), generator=g).cuda(), attention_mask=torch.ones(1, T).cuda(),
              token_type_ids=torch.zeros(1, T, dtype=torch.long).cuda(),
              visual_feats=torch.randn(1, I, 2048, generator=g).cuda(), visual_pos=torch.rand(1, I, 4, generator=g).cuda())
cam_i, cam_t = torch.rand(I, generator=g).cuda(), torch.rand(T, generator=g).cuda()
pert = lp.LxmertPerturbation(model)


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


@torch.no_grad()
def sequential_image():
    for step in lp.PERT_STEPS:
        idx = cam_i.topk(k=int((1 - step) * I), dim=-1).indices.cpu().numpy()       # host round trip, as the reference
        model(input_ids=inputs["input_ids"], attention_mask=inputs["attention_mask"],
              token_type_ids=inputs["token_type_ids"], visual_feats=inputs["visual_feats"][:, idx],
              visual_pos=inputs["visual_pos"][:, idx]).question_answering_score.argmax().item()


print("image test, 9 sequential forwards : %.2f ms / sample" % timed(sequential_image))
print("image test, batched               : %.2f ms / sample" % timed(lambda: pert.perturbation_image(inputs, cam_i)))
print("text  test, batched               : %.2f ms / sample" % timed(lambda: pert.perturbation_text(inputs, cam_t)))

# relevancy generation itself (config 4 sizes: T = 14 question tokens, I = 36 regions, 9 + 5 + 5 layers)
import types  # noqa: E402

from transformer_mm_explainability_amd import lxmert_explainability as le  # noqa: E402

usage = types.SimpleNamespace(model=model, text_len=T, image_boxes_len=I, forward=lambda item: model(**inputs))
gen = le.GeneratorOurs(usage)
print("GeneratorOurs.generate_ours (fused)   : %.2f ms / sample" % timed(lambda: gen.generate_ours(None, use_lrp=False)))
with torch.no_grad():
    print("forward only                          : %.2f ms / sample" % timed(lambda: model(**inputs)))

for B in (8, 32):
    gb = torch.Generator().manual_seed(2)
    batch = dict(input_ids=torch.randint(1, 30000, (B, T), generator=gb).cuda(), attention_mask=torch.ones(B, T).cuda(),
                 token_type_ids=torch.zeros(B, T, dtype=torch.long).cuda(),
                 visual_feats=torch.randn(B, I, 2048, generator=gb).cuda(), visual_pos=torch.rand(B, I, 4, generator=gb).cuda())
    ms = timed(lambda: gen.generate_ours_batch(batch), n=5)
    run = le.GraphedGenerateOursBatch(model, batch)
    mg = timed(lambda: run(batch), n=10)
    print("B=%-2d generate_ours_batch eager %.2f ms | hipGraph replay (+ one deferred diag check) %.2f ms = %.3f ms/sample"
          % (B, ms, mg, mg / B))
    del run
    cams = torch.rand(B, I, generator=gb).cuda()
    mp = timed(lambda: pert.perturbation_image(batch, cams), n=5)
    print("B=%-2d generate_ours_batch %.2f ms (%.2f ms/sample) | image test %.2f ms (%.2f ms/sample) -> %.0f samples/s for both"
          % (B, ms, ms / B, mp, mp / B, B / (ms + mp) * 1e3))
