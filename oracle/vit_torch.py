"""CPU oracle (torch fp32) for the ViT path: timm-layout ViT with hooked attention + the notebook's ``generate_relevance``.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED for the model body: the notebook's model class (``baselines/ViT/ViT_new.py``)
is in another repository (``hila-chefer/Transformer-Explainability``, cloned unpinned, not in /root/reference) and no
reference test pins its outputs.  Restated from its published algorithm (timm ViT: pre-LN blocks, ``softmax(q k^T * scale)``,
exact GELU, class token) and anchored on the notebook's call sites: ``model(x, register_hook=True)``,
``blk.attn.get_attention_map()`` / ``get_attn_gradients()`` (``Transformer_MM_explainability_ViT.ipynb`` cell 7:15,27-33).
The rule chain it feeds IS pinned (``tests/golden/vit_chain.npz``, made by exec'ing the notebook cell).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def forward(sd, x, heads):
    """timm ViT forward from a state dict; returns ``(logits, probs[list of [B,H,N,N], graph-attached])``."""
    p = sd["patch_embed.proj.weight"].shape[-1]
    x = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=p).flatten(2).transpose(1, 2)
    x = torch.cat([sd["cls_token"].expand(x.shape[0], -1, -1), x], dim=1) + sd["pos_embed"]
    B, N, E = x.shape
    d = E // heads
    probs = []
    depth = len([k for k in sd if k.endswith(".attn.qkv.weight")])
    for l in range(depth):
        pre = "blocks.%d." % l
        h = F.layer_norm(x, (E,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], 1e-6)
        qkv = F.linear(h, sd[pre + "attn.qkv.weight"], sd[pre + "attn.qkv.bias"]).reshape(B, N, 3, heads, d).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = ((q @ k.transpose(-2, -1)) * (d ** -0.5)).softmax(dim=-1)
        probs.append(attn)                                             # save_attention_map + register_hook
        o = (attn @ v).transpose(1, 2).reshape(B, N, E)
        x = x + F.linear(o, sd[pre + "attn.proj.weight"], sd[pre + "attn.proj.bias"])
        h = F.layer_norm(x, (E,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], 1e-6)
        h = F.gelu(F.linear(h, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"]))
        x = x + F.linear(h, sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])
    x = F.layer_norm(x, (E,), sd["norm.weight"], sd["norm.bias"], 1e-6)
    return F.linear(x[:, 0], sd["head.weight"], sd["head.bias"]), probs


def generate_relevance(sd, x, heads, index=None):
    """Notebook cell 7:14-34: one backward, then ``R += avg_heads(attn, grad) @ R`` over the blocks; ``R[0, 1:]``."""
    sd = {k: v.detach().float().clone().requires_grad_(True) for k, v in sd.items()}
    logits, probs = forward(sd, x, heads)
    if index is None:
        index = int(np.argmax(logits.detach().numpy(), axis=-1)[0])
    one_hot = torch.zeros_like(logits)
    one_hot[0, index] = 1
    grads = torch.autograd.grad(torch.sum(one_hot * logits), probs)
    n = probs[0].shape[-1]
    R = torch.eye(n)
    for attn, grad in zip(probs, grads):
        cam = (grad.reshape(-1, n, n) * attn.detach().reshape(-1, n, n)).clamp(min=0).mean(dim=0)
        R = R + cam @ R
    return R[0, 1:], logits.detach()
