"""ViT relevancy -- ``Transformer_MM_explainability_ViT.ipynb`` cell 7 (``avg_heads``, 2-argument
``apply_self_attention_rules``, ``generate_relevance(model, input, index=None)``) on the HIP chain kernel.

``model`` is duck-typed as in the notebook (the class lives in the external ``Transformer-Explainability`` repo, not in
the reference tree): ``model(input, register_hook=True) -> logits [1, C]``, ``model.blocks[i].attn.get_attention_map()``
and ``.get_attn_gradients()`` -> ``[1, H, N, N]`` device tensors.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops, rules

avg_heads = rules.avg_heads
apply_self_attention_rules = rules.apply_self_attention_rules_vit


def generate_relevance(model, input, index=None):
    """Notebook cell 7:14-34 -> ``R[0, 1:]`` (``[N-1]``).  All blocks' rule 5+6 updates run in one kernel launch."""
    output = model(input, register_hook=True)
    if index is None:
        index = np.argmax(output.cpu().data.numpy(), axis=-1)
    one_hot = torch.zeros_like(output)
    one_hot[0, index] = 1
    loss = torch.sum(one_hot * output)
    model.zero_grad()
    loss.backward(retain_graph=True)
    attn = [blk.attn.get_attention_map().detach() for blk in model.blocks]
    grad = [blk.attn.get_attn_gradients().detach() for blk in model.blocks]
    return ops.relevancy_chain_row(attn, grad, 1, 0)[0, 1:]          # row 0 of R only: carried as a row vector (N = 197 > 128)
