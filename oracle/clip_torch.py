"""CPU oracle (torch fp32) of the reference's CLIP hook path + notebook ``interpret``.

TEST INFRASTRUCTURE ONLY (see ``oracle/relevancy_np.py`` header for who may import this).  Self-contained:
takes a plain ``state_dict`` with the reference's parameter names and runs stock torch CPU ops; it does not
import the product package.

Restates, with citations:
  * the hooked attention core ``softmax((q*scale) k^T + mask)`` whose output is captured and whose gradient
    is captured by a tensor hook -- CLIP/clip/auxilary.py:153,225-252 and CLIP/clip/model.py:181-198
  * the ViT / text towers -- CLIP/clip/model.py:229-246, 343-378
  * ``interpret`` exactly as the notebook runs it: one ``torch.autograd.grad(one_hot, [probs_l], retain_graph=True)``
    PER LAYER (CLIP_explainability.ipynb cell 6:22-32, 45-55), i.e. L partial backward passes -- this is the cost
    profile ``bench.py``'s ``cpu_baseline`` measures.
Pinned by ``tests/test_oracle_golden.py::test_clip_torch_oracle`` against ``tests/golden/clip_tiny.npz`` (produced by
the reference's own model + notebook cell).

Half precision (``prepare_state_dict(..., dtype=torch.float16)``): the reference's own GPU mode -- ``convert_weights``
(CLIP/clip/model.py:381-402) rounds the Linear / Conv / attention / projection parameters to fp16, LayerNorm computes in fp32 and
casts back (model.py:157-164), activations, gradients and the R chain (created in the dtype of the probabilities, notebook cell
6:20,43) are fp16.  Pinned by ``test_clip_torch_oracle_fp16`` against ``tests/golden/clip_tiny_fp16.npz`` (the reference's model
after ``convert_weights``, run on the CPU).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _layer_norm(x, weight, bias):
    """model.py:157-164: fp32 statistics whatever the activation dtype, cast back."""
    return F.layer_norm(x.float(), (x.shape[-1],), weight, bias).to(x.dtype)


def _block(sd, prefix, x, heads, mask, captured):
    """One pre-LN residual block on ``x [B, N, E]``; appends the (graph-attached) probabilities to ``captured``."""
    B, N, E = x.shape
    d = E // heads
    h = _layer_norm(x, sd[prefix + "ln_1.weight"], sd[prefix + "ln_1.bias"])
    qkv = F.linear(h, sd[prefix + "attn.in_proj_weight"], sd[prefix + "attn.in_proj_bias"])
    q, k, v = qkv.chunk(3, dim=-1)
    q = q * (float(d) ** -0.5)                                     # auxilary.py:153
    q = q.view(B, N, heads, d).permute(0, 2, 1, 3).reshape(B * heads, N, d)
    k = k.view(B, N, heads, d).permute(0, 2, 1, 3).reshape(B * heads, N, d)
    v = v.view(B, N, heads, d).permute(0, 2, 1, 3).reshape(B * heads, N, d)
    w = torch.bmm(q, k.transpose(1, 2))                            # auxilary.py:225
    if mask is not None:
        w = w + mask                                               # auxilary.py:232
    w = F.softmax(w, dim=-1)                                       # auxilary.py:243
    captured.append(w)                                             # forward hook, auxilary.py:248
    o = torch.bmm(w, v).view(B, heads, N, d).permute(0, 2, 1, 3).reshape(B, N, E)
    x = x + F.linear(o, sd[prefix + "attn.out_proj.weight"], sd[prefix + "attn.out_proj.bias"])
    h = _layer_norm(x, sd[prefix + "ln_2.weight"], sd[prefix + "ln_2.bias"])
    h = F.linear(h, sd[prefix + "mlp.c_fc.weight"], sd[prefix + "mlp.c_fc.bias"])
    h = h * torch.sigmoid(1.702 * h)
    return x + F.linear(h, sd[prefix + "mlp.c_proj.weight"], sd[prefix + "mlp.c_proj.bias"])


def _n_layers(sd, prefix):
    return len([k for k in sd if k.startswith(prefix) and k.endswith(".attn.in_proj_weight")])


def forward(sd, images, texts):
    """Returns ``(logits_per_image, img_probs[list of L], txt_probs[list of L])``; probs ``[B*H, N, N]``."""
    width = sd["visual.conv1.weight"].shape[0]
    patch = sd["visual.conv1.weight"].shape[-1]
    dt = sd["visual.conv1.weight"].dtype                           # ``CLIP.dtype`` (model.py:343-345)
    x = F.conv2d(images.to(dt), sd["visual.conv1.weight"], stride=patch).flatten(2).transpose(1, 2)
    cls = sd["visual.class_embedding"].to(dt).expand(x.shape[0], 1, -1)            # model.py:233
    x = torch.cat([cls, x], dim=1) + sd["visual.positional_embedding"].to(dt)      # model.py:234
    x = _layer_norm(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])
    img_probs = []
    for l in range(_n_layers(sd, "visual.transformer.resblocks.")):
        x = _block(sd, "visual.transformer.resblocks.%d." % l, x, width // 64, None, img_probs)
    x = _layer_norm(x[:, 0, :], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"])
    image_features = x @ sd["visual.proj"]

    tw = sd["ln_final.weight"].shape[0]
    ctx = sd["positional_embedding"].shape[0]
    mask = torch.full((ctx, ctx), float("-inf")).triu_(1).to(dt)                   # model.py:177
    t = F.embedding(texts, sd["token_embedding.weight"]).to(dt) + sd["positional_embedding"].to(dt)   # model.py:350-352
    txt_probs = []
    heads = sd["__text_heads__"]
    for l in range(_n_layers(sd, "transformer.resblocks.")):
        t = _block(sd, "transformer.resblocks.%d." % l, t, heads, mask, txt_probs)
    t = _layer_norm(t, sd["ln_final.weight"], sd["ln_final.bias"])
    text_features = t[torch.arange(t.shape[0]), texts.argmax(dim=-1)] @ sd["text_projection"]

    image_features = image_features / image_features.norm(dim=-1, keepdim=True)
    text_features = text_features / text_features.norm(dim=-1, keepdim=True)
    logits_per_image = sd["logit_scale"].exp() * image_features @ text_features.t()
    return logits_per_image, img_probs, txt_probs


def _chain_reference_style(one_hot, probs, batch_size, start_layer, timings=None):
    import time
    n = probs[0].shape[-1]
    R = torch.eye(n, dtype=probs[0].dtype).unsqueeze(0).expand(batch_size, n, n)
    for i, p in enumerate(probs):
        if i < start_layer:
            continue
        t0 = time.perf_counter()
        grad = torch.autograd.grad(one_hot, [p], retain_graph=True)[0].detach()   # cell 6:25 -- per layer
        if timings is not None:
            timings["backward_s"] = timings.get("backward_s", 0.0) + time.perf_counter() - t0
        cam = p.detach().reshape(-1, n, n)
        cam = (grad.reshape(-1, n, n) * cam).reshape(batch_size, -1, n, n)
        cam = cam.clamp(min=0).mean(dim=1)
        R = R + torch.bmm(cam, R)
    return R


def chain_half(attn_layers, grad_layers, batch_size):
    """The rule of cell 6:22-32 on given fp16 slabs ``[B*H, N, N]`` (no autograd): what the reference's chain computes once the
    probabilities and their gradients are fp16 tensors -- every op returns fp16 (fp32 arithmetic inside, one rounding)."""
    n = attn_layers[0].shape[-1]
    R = torch.eye(n, dtype=torch.float16).unsqueeze(0).expand(batch_size, n, n)
    for a, g in zip(attn_layers, grad_layers):
        a, g = a.to(torch.float16), g.to(torch.float16)
        cam = (g.reshape(-1, n, n) * a.reshape(-1, n, n)).reshape(batch_size, -1, n, n)
        cam = cam.clamp(min=0).mean(dim=1)
        R = R + torch.bmm(cam, R)
    return R


def _converted_by_convert_weights(name):
    """Which parameters ``convert_weights`` (model.py:381-402) rounds to fp16: nn.Linear / nn.Conv2d weights and biases, the
    attention module's packed projection, ``text_projection`` and the visual ``proj``.  LayerNorm parameters, the embeddings
    and ``logit_scale`` stay fp32 (the embeddings are cast at their use, model.py:233-234, 350-352)."""
    return (name.endswith(("conv1.weight", "in_proj_weight", "in_proj_bias", "out_proj.weight", "out_proj.bias",
                           "c_fc.weight", "c_fc.bias", "c_proj.weight", "c_proj.bias"))
            or name in ("text_projection", "visual.proj"))


def prepare_state_dict(state_dict, text_heads, dtype=torch.float32):
    """Leaf tensors that require grad (like ``nn.Parameter``: the reference never freezes them) + text heads.
    ``dtype=torch.float16``: the parameters ``convert_weights`` converts are rounded to fp16, the others stay fp32."""
    sd = {}
    for k, v in state_dict.items():
        if torch.is_tensor(v) and v.is_floating_point():
            t = v.detach().float().clone()
            if dtype != torch.float32 and _converted_by_convert_weights(k):
                t = t.to(dtype)
            sd[k] = t.requires_grad_(True)
    sd["__text_heads__"] = int(text_heads)
    return sd


def interpret(sd, image, texts, start_layer=-1, start_layer_text=-1, timings=None):
    """Notebook ``interpret`` (cell 6): returns ``(R_text [B,Nt,Nt], image_relevance [B,Ni-1])``."""
    import time
    t0 = time.perf_counter()
    batch_size = texts.shape[0]
    images = image.repeat(batch_size, 1, 1, 1)
    logits_per_image, img_probs, txt_probs = forward(sd, images, texts)
    one_hot = torch.sum(torch.eye(batch_size) * logits_per_image)            # cell 6:8-10 (fp32 eye: the product promotes)
    t1 = time.perf_counter()
    if start_layer == -1:
        start_layer = len(img_probs) - 1
    if start_layer_text == -1:
        start_layer_text = len(txt_probs) - 1
    split = {} if timings is not None else None
    R = _chain_reference_style(one_hot, img_probs, batch_size, start_layer, split)
    R_text = _chain_reference_style(one_hot, txt_probs, batch_size, start_layer_text, split)
    t2 = time.perf_counter()
    if timings is not None:      # forward / the per-layer partial backwards / the rule chain, separately
        bwd = split.get("backward_s", 0.0)
        timings.update(forward_s=t1 - t0, backward_s=bwd, rules_s=t2 - t1 - bwd, backward_and_rules_s=t2 - t1)
    return R_text, R[:, 0, 1:]


def capture_all(sd, image, texts):
    """ONE backward through every captured tensor: ``(logits, img_probs, img_grads, txt_probs, txt_grads)``."""
    batch_size = texts.shape[0]
    logits, img_probs, txt_probs = forward(sd, image.repeat(batch_size, 1, 1, 1), texts)
    one_hot = torch.sum(torch.eye(batch_size) * logits)
    grads = torch.autograd.grad(one_hot, img_probs + txt_probs)
    L = len(img_probs)
    return (logits.detach(), [p.detach() for p in img_probs], list(grads[:L]),
            [p.detach() for p in txt_probs], list(grads[L:]))
