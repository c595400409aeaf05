"""ViT (timm layout) whose attention runs on the HIP capture op -- the body behind ``generate_relevance``
(``Transformer_MM_explainability_ViT.ipynb`` cells 7-8, BASELINE config "ViT-B/16 single-image class-index relevancy").

The notebook's model class (``baselines/ViT/ViT_new.py``) lives in a different repository
(``hila-chefer/Transformer-Explainability``, cloned unpinned by the notebook, NOT in the reference tree), so this file is
written against the surface the notebook uses and the timm ViT parameter layout that class loads:

  * ``model(x, register_hook=True) -> logits``; ``model.blocks[i].attn.get_attention_map()`` /
    ``.get_attn_gradients()`` -> ``[B, H, N, N]`` (views of the capture slabs here; ``register_hook`` is accepted and
    ignored: capture is always on, there are no hooks)
  * parameters: ``patch_embed.proj``, ``cls_token``, ``pos_embed``, ``blocks.{i}.norm1 / attn.qkv / attn.proj / norm2 /
    mlp.fc1 / mlp.fc2``, ``norm``, ``head`` (a timm ``vit_base_patch16_224`` state dict loads unchanged)
  * attention: ``softmax((q k^T) * scale)`` (scale AFTER the product, like ViT_new / timm) -> MMX_SCALE_SCORES with
    divisor ``1/scale``; exact GELU (erf).

``generate_relevance_multi`` is SURVEY.md section 8f row 1 (batched-target backward): the notebook explains two classes
of one image with two full forward+backward passes (cell 9:12-16); here the forward runs ONCE and the K one-hot
backward passes run as one batch-K backward over the shared activations (same scheme as CLIP's shared image tower).
"""
from __future__ import annotations


import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops
from .capture import CaptureBuffers, attention_capture_packed


class Attention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias=True):
        super().__init__()
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.attention_map = None
        self.attn_gradients = None

    # the surface ViT_new exposes to the notebook (cell 7:30-31)
    def get_attention_map(self):
        return self.attention_map

    def get_attn_gradients(self):
        return self.attn_gradients

    def save_attention_map(self, attention_map):
        self.attention_map = attention_map

    def save_attn_gradients(self, attn_gradients):
        self.attn_gradients = attn_gradients

    def forward(self, x, probs, grads):
        B, N, C = x.shape
        qkv = self.qkv(x).view(B, N, 3, self.num_heads, self.head_dim)
        self.attention_map, self.attn_gradients = probs, grads
        o = attention_capture_packed(qkv, probs, grads, 1.0 / self.scale, scale_mode=_lib.SCALE_SCORES)
        return self.proj(o.view(B, N, C))


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=True):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, num_heads, qkv_bias)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x, probs, grads):
        x = x + self.attn(self.norm1(x), probs, grads)
        return x + self.mlp(self.norm2(x))


class PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        self.patch_size = patch_size
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        """kernel == stride == patch: the conv is a GEMM over flattened patches (keeps MIOpen off the path)."""
        B, C, H, W = x.shape
        p = self.patch_size
        gh, gw = H // p, W // p
        patches = x.reshape(B, C, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * p * p)
        return F.linear(patches, self.proj.weight.reshape(self.proj.out_channels, -1), self.proj.bias)


class VisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4.0, qkv_bias=True):
        super().__init__()
        self.num_classes, self.embed_dim, self.depth, self.num_heads = num_classes, embed_dim, depth, num_heads
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.head = nn.Linear(embed_dim, num_classes)
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        self.apply(self._init_weights)
        self.buffers_ = None

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.LayerNorm):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)

    def _gemm(self, x, weight):
        """``x @ weight`` of the hand-written backward in ``backward_gemm_dtype`` (default fp32; ``torch.bfloat16`` =
        bf16 MFMA with fp32 accumulation and results, see ``clip_model.Transformer._gemm``)."""
        return ops.backward_gemm(x, weight, getattr(self, "backward_gemm_dtype", torch.float32))

    def _slabs(self, batch, n_tokens, device, shared=False):
        b = self.buffers_
        dtype = getattr(self, "capture_dtype", torch.float32)   # torch.float16 / bfloat16: half-size slabs (N >= ~128)
        if b is None or not b.matches(self.depth, batch, self.num_heads, n_tokens, n_tokens, device, shared, dtype):
            self.buffers_ = b = CaptureBuffers(self.depth, batch, self.num_heads, n_tokens, n_tokens, device=device,
                                               shared_probs=shared, dtype=dtype)
        return b

    def _embed(self, x):
        x = self.patch_embed(x)
        return torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], dim=1) + self.pos_embed

    def forward(self, x, register_hook=False):
        if not x.is_cuda:
            raise _lib.MMXError("the ViT body runs its attention on the HIP capture op: move model and input to the GPU")
        x = self._embed(x)
        buf = self._slabs(x.shape[0], x.shape[1], x.device)
        for l, blk in enumerate(self.blocks):
            x = blk(x, buf.probs[l], buf.grads[l])
        return self.head(self.norm(x)[:, 0])

    # ---------------------------------------------------------------- shared forward, batch-K backward (section 8f row 1)
    @torch.no_grad()
    def forward_shared(self, x, n_targets):
        """One forward at batch 1 that keeps what the batched backward needs.  Returns ``(logits [1, C], state)``."""
        x = self._embed(x)
        N, E = x.shape[1], x.shape[2]
        buf = self._slabs(n_targets, N, x.device, shared=True)
        tape = []
        # every LayerNorm but the first is fused with the residual add that produces its input (ops.add_layernorm: one pass
        # writes the sum, the normalised rows and their statistics)
        first = self.blocks[0].norm1
        _, h1, mean1, rstd1 = ops.add_layernorm(x, None, first.weight, first.bias, first.eps)
        for l, blk in enumerate(self.blocks):
            at = blk.attn
            qkv = at.qkv(h1).view(1, N, 3, at.num_heads, at.head_dim)
            o = ops.attn_capture_fwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], buf.probs[l], 1.0 / at.scale,
                                     _lib.SCALE_SCORES, None, layout="bnhd")
            x1, h2, mean2, rstd2 = ops.add_layernorm(x, at.proj(o.view(1, N, E)), blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
            m = blk.mlp.fc1(h2)
            mlp_out = blk.mlp.fc2(F.gelu(m))
            tape.append((x, mean1, rstd1, qkv, x1, mean2, rstd2, m, o))
            at.attention_map, at.attn_gradients = buf.probs[l], buf.grads[l]
            nxt = self.blocks[l + 1].norm1 if l + 1 < self.depth else self.norm
            x, h1, mean1, rstd1 = ops.add_layernorm(x1, mlp_out, nxt.weight, nxt.bias, nxt.eps)
        return self.head(h1[:, 0]), (tape, x, mean1, rstd1)

    @torch.no_grad()
    def backward_shared(self, state, d_logits, on_layer_done=None):
        """``d_logits [K, C]``: K upstream gradients over the ONE forward; fills ``grads`` of every block (batch K).
        ``on_layer_done(l)``: called right after block l's gradient slab is complete on the current stream (the backward runs
        top-down -- the order a relevancy ROW is carried in -- so a caller can apply layer l's rule beside the rest of it)."""
        tape, x_last, mean, rstd = state
        K = d_logits.shape[0]
        N, E = x_last.shape[1], x_last.shape[2]
        d_f = torch.zeros(K, N, E, dtype=torch.float32, device=d_logits.device)
        d_f[:, 0, :] = torch.matmul(d_logits, self.head.weight)
        dx = ops.layernorm_bwd_add(d_f, x_last, mean, rstd, self.norm.weight)
        buf = self.buffers_
        for l in range(self.depth - 1, -1, -1):
            blk = self.blocks[l]
            at = blk.attn
            x, mean1, rstd1, qkv, x1, mean2, rstd2, m, o_fwd = tape[l]
            d_a = self._gemm(dx, blk.mlp.fc2.weight)
            # d_a * GELU'(m), exact (erf) form, in ONE library kernel (m [1, N, 4E] is the shared pre-activation: broadcast)
            d_m = torch.ops.aten.gelu_backward(d_a, m.expand_as(d_a))
            d_h2 = self._gemm(d_m, blk.mlp.fc1.weight)
            d_x1 = ops.layernorm_bwd_add(d_h2, x1, mean2, rstd2, blk.norm2.weight, dx)
            d_o = self._gemm(d_x1, at.proj.weight).view(K, N, at.num_heads, at.head_dim)
            need = l > 0
            dqkv = torch.empty(K, N, 3, at.num_heads, at.head_dim, dtype=torch.float32, device=dx.device) if need else None
            out = (dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2]) if need else None
            ops.attn_capture_bwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], buf.probs[l], d_o, buf.grads[l], 1.0 / at.scale,
                                 _lib.SCALE_SCORES, need_dqkv=need, layout="bnhd", out=out, batch=K, o=o_fwd)
            if on_layer_done is not None:
                on_layer_done(l)
            if not need:
                break
            d_h1 = self._gemm(dqkv.view(K, N, 3 * E), at.qkv.weight)
            dx = ops.layernorm_bwd_add(d_h1, x, mean1, rstd1, blk.norm1.weight, d_x1)


def vit_base_patch16_224(num_classes=1000, **kw):
    """Architecture of the notebook's ``vit(pretrained=True)`` (cell 8:20); weights are NOT downloaded here."""
    return VisionTransformer(img_size=224, patch_size=16, embed_dim=768, depth=12, num_heads=12, num_classes=num_classes, **kw)


def generate_relevance_multi(model, input, indices=None, top_k=None):
    """Relevancy maps of ONE image for K class indices with one forward (section 8f row 1).

    Equivalent to ``[generate_relevance(model, input, index=k) for k in indices]`` (notebook cell 9:12-16 runs those as
    separate full passes).  ``indices=None``: the ``top_k`` highest-scoring classes, picked on the device (the
    notebook's default ``index=None`` is ``top_k=1``).  Returns ``[K, N-1]``.
    """
    if indices is None:
        K = 1 if top_k is None else int(top_k)
        idx = None
    else:
        idx = torch.as_tensor(indices, device=input.device).reshape(-1)
        K = idx.numel()
    logits, state = model.forward_shared(input, K)
    if idx is None:
        idx = logits[0].topk(K).indices
    d_logits = torch.zeros(K, logits.shape[-1], dtype=torch.float32, device=input.device)
    d_logits.scatter_(1, idx.reshape(K, 1), 1.0)                     # one one-hot per target (capturable: no host sync)
    buf = model.buffers_
    N = buf.probs[0].shape[-1]
    if N <= 128:        # the one-launch fused chain (all layers, R in registers) is faster there; the row is picked afterwards
        model.backward_shared(state, d_logits)
        row = ops.relevancy_chain_row([buf.probs[l] for l in range(model.depth)], [buf.grads[l] for l in range(model.depth)],
                                      K, 0, shared_attn=K > 1)
        return row[:, 1:]
    # R[:, 0, :] carried as a ROW vector from the top layer down, x <- x + x . A_bar_l (no 197^3 products): the backward visits the
    # layers in that same order, so layer l's head average + mat-vec start as soon as its gradient slab is complete, on a side
    # stream beside the rest of the backward
    main = torch.cuda.current_stream()
    # (eager calls only: THIS fork pattern captured into a hipGraph -- 12 re-waits of the side stream on the main one, three
    # kernels each -- replayed fine in a fresh process and segfaulted inside hipGraphLaunch once CLIP graphs existed in the
    # process (bench.py's leg order; bisected with MMX_DEBUG_NO_SIDE).  Under capture the rule kernels stay on the main stream.)
    side = main if torch.cuda.is_current_stream_capturing() else ops.side_stream(input.device)
    x = torch.zeros(K, N, dtype=torch.float32, device=input.device)
    x[:, 0] = 1.0
    row = [x]

    def layer_rule(l):
        if side is main:
            row[0] = ops.avg_heads_vecmat(row[0], buf.probs[l], buf.grads[l], batch_size=K, shared_attn=K > 1)
            return
        side.wait_stream(main)
        with torch.cuda.stream(side):
            row[0] = ops.avg_heads_vecmat(row[0], buf.probs[l], buf.grads[l], batch_size=K, shared_attn=K > 1)

    model.backward_shared(state, d_logits, layer_rule)
    if side is not main:
        main.wait_stream(side)
        row[0].record_stream(main)
    return row[0][:, 1:]


class GraphedRelevance:
    """``generate_relevance_multi`` captured once into a hipGraph and replayed.  At batch 1 the pass is ~700 launches
    of a few microseconds each, i.e. bound by the Python + launch path on the host; a replay is one ``hipGraphLaunch``.

        run = GraphedRelevance(model, image, indices=[243, 282])      # or top_k=1: arg-max class, chosen on the device
        maps = run(image2, [1, 7])                                    # [K, N-1]; same values as the eager function

    K and the image shape are fixed at construction; the returned tensor is the graph's output buffer (overwritten by
    the next call, ``.clone()`` to keep it).
    """

    def __init__(self, model, input, indices=None, top_k=None, warmup=3):
        self.input = input.clone()
        self.indices = None if indices is None else torch.as_tensor(indices, device=input.device).reshape(-1).clone()
        call = lambda: generate_relevance_multi(model, self.input, self.indices, top_k)   # noqa: E731
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                call()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with ops.graph_capture(self.graph):
            self.output = call()

    def __call__(self, input=None, indices=None):
        if input is not None:
            self.input.copy_(input)
        if indices is not None:
            if self.indices is None:
                raise ValueError("this graph picks its classes itself (top_k mode)")
            self.indices.copy_(torch.as_tensor(indices, device=self.indices.device).reshape(-1))
        self.graph.replay()
        return self.output
