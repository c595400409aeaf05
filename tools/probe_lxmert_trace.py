"""Probe (GPU box, under rocprofv3 --kernel-trace --stats): the batched LXMERT explain pass (B = 32, T = 14, I = 36), eager, a few times."""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformer_mm_explainability_amd import lxmert_explainability as le  # noqa: E402
from transformer_mm_explainability_amd import lxmert_model as lm  # noqa: E402

torch.manual_seed(0)
model = lm.LxmertForQuestionAnswering(lm.LxmertConfig()).cuda().eval()
B, T, I = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 14, 36
gb = torch.Generator().manual_seed(2)
batch = dict(input_ids=torch.randint(1, 30000, (B, T), generator=gb).cuda(), attention_mask=torch.ones(B, T).cuda(),
             token_type_ids=torch.zeros(B, T, dtype=torch.long).cuda(),
             visual_feats=torch.randn(B, I, 2048, generator=gb).cuda(), visual_pos=torch.rand(B, I, 4, generator=gb).cuda())
gen = le.GeneratorOurs(types.SimpleNamespace(model=model))
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 5):
    gen.generate_ours_batch(batch, check_diag="defer")
torch.cuda.synchronize()
