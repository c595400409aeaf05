"""CLIP (ViT image tower + causal text tower) whose attention runs on the HIP capture op.

Role in this repo: the transformer body that FEEDS the relevancy engine.  Per BASELINE.json's north_star the
body's GEMMs / LayerNorms stay on PyTorch-ROCm; only the attention core is ours, writing P and dP into
preallocated slabs (``capture.CaptureBuffers``).  Parameter names and shapes are those of the reference's
``CLIP/clip/model.py`` (``visual.transformer.resblocks.{i}.attn.in_proj_weight`` ...), so a reference /
OpenAI ViT state dict loads with ``load_state_dict`` unchanged; the surface the explainability code touches
is kept too (reference file:line in the docstrings):

  * ``model.visual.transformer.resblocks`` / ``model.transformer.resblocks`` iterable of blocks
  * ``blk.attn_probs`` / ``blk.attn_grad`` -> ``[B*H, N, N]`` (CLIP/clip/model.py:181-193), here views of the slabs
  * ``model(image, text) -> (logits_per_image, logits_per_text)`` (model.py:364-378)

Design differences (MI355X-first, not a port): activations are batch-first ``[B, N, E]`` end to end (the
reference permutes to LND and back), q/k/v are strided views of the packed in-projection output consumed
in place by the kernel, the head-concatenated output is written by the kernel in the layout ``out_proj`` reads,
and ``capture_only`` mode cuts the autograd graph below the first block so ONE backward yields every layer's
dP without weight or input gradients.  ModifiedResNet towers (RN50 etc.) have no attention stack to explain
(reference notebooks use ViT-B/32) and are not built.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops
from .capture import CaptureBuffers, attention_capture_packed


class QuickGELU(nn.Module):
    def forward(self, x):
        if x.is_cuda and x.dtype == torch.float32:
            return ops.quick_gelu(x)            # one fused HBM pass forward, one backward
        return x * torch.sigmoid(1.702 * x)     # parameter-only / CPU use of the module (no attention can run there)


class _OutProj(nn.Linear):
    """``attn.out_proj`` (name kept for state-dict compatibility)."""


class CapturedSelfAttention(nn.Module):
    """Packed-projection multi-head self-attention.  Parameters: ``in_proj_weight [3E, E]``, ``in_proj_bias``,
    ``out_proj`` -- the ``nn.MultiheadAttention`` naming the reference's ``MultiheadAttention`` uses
    (CLIP/clip/auxilary.py:265-356)."""

    def __init__(self, embed_dim, num_heads):
        super().__init__()
        if embed_dim % num_heads:
            raise ValueError("embed_dim must be divisible by num_heads")
        self.embed_dim, self.num_heads, self.head_dim = embed_dim, num_heads, embed_dim // num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = _OutProj(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)

    def forward(self, x, probs_slab, grads_slab, mask=None, need_dqkv=True, grad_hook=None):
        B, N, E = x.shape
        qkv = F.linear(x, self.in_proj_weight, self.in_proj_bias).view(B, N, 3, self.num_heads, self.head_dim)
        o = attention_capture_packed(qkv, probs_slab, grads_slab, self.head_dim ** -0.5, mask=mask,
                                     scale_mode=_lib.SCALE_Q_FIRST, need_dqkv=need_dqkv, grad_hook=grad_hook)
        return self.out_proj(o.view(B, N, E))


class ResidualAttentionBlock(nn.Module):
    """Pre-LN block (CLIP/clip/model.py:167-198).  ``attn_probs`` / ``attn_grad`` are slab views."""

    def __init__(self, d_model, n_head, attn_mask=None):
        super().__init__()
        self.attn = CapturedSelfAttention(d_model, n_head)
        self.ln_1 = nn.LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([
            ("c_fc", nn.Linear(d_model, d_model * 4)),
            ("gelu", QuickGELU()),
            ("c_proj", nn.Linear(d_model * 4, d_model)),
        ]))
        self.ln_2 = nn.LayerNorm(d_model)
        self.attn_mask = attn_mask
        self.attn_probs = None
        self.attn_grad = None

    # kept for API parity with the reference block (model.py:184-188)
    def set_attn_probs(self, attn_probs):
        self.attn_probs = attn_probs

    def set_attn_grad(self, attn_grad):
        self.attn_grad = attn_grad

    def forward(self, x, buffers, layer, need_dqkv=True):
        if self.attn_mask is not None and (self.attn_mask.device != x.device):
            self.attn_mask = self.attn_mask.to(device=x.device, dtype=torch.float32)
        probs, grads = buffers.probs[layer], buffers.grads[layer]
        self.attn_probs = buffers.layer_probs(layer)
        self.attn_grad = buffers.layer_grads(layer)  # valid once backward has run
        mask = self.attn_mask
        if mask is not None and mask.shape[-1] != x.shape[1]:
            mask = mask[: x.shape[1], : x.shape[1]].contiguous()   # trimmed (padding-free) text batch
        x = x + self.attn(self.ln_1(x), probs, grads, mask=mask, need_dqkv=need_dqkv)
        x = x + self.mlp(self.ln_2(x))
        return x


class Transformer(nn.Module):
    def __init__(self, width, layers, heads, attn_mask=None):
        super().__init__()
        self.width, self.layers, self.heads = width, layers, heads
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads, attn_mask) for _ in range(layers)])
        self.buffers = None
        # element type of the capture slabs: torch.float16 / torch.bfloat16 halve the resident bytes of a long-sequence
        # tower (ViT-L/14@336: 577 tokens); only the streaming attention kernels write them (capture op, ABI note)
        self.capture_dtype = torch.float32
        # dtype of the input-gradient GEMMs of ``backward_shared`` (the bulk of a long-sequence step: ViT-L/14@336 at
        # batch 128 is ~45 TFLOP of them).  torch.bfloat16 runs them on the bf16 MFMA with fp32 accumulation and fp32
        # results; everything else (forward, LayerNorm, attention, the relevancy rules) stays fp32.
        self.backward_gemm_dtype = torch.float32
        # A bf16 BODY (BASELINE config 5; the reference converts the weights, CLIP/clip/model.py:381-402, and runs the
        # whole tower in half precision): ``forward_gemm_dtype`` rounds the operands of the tape forward's GEMMs to bf16
        # (fp32 accumulate, fp32 activations between the GEMMs), ``attention_mma_bf16`` runs the attention products of
        # the long-sequence streaming kernels on the bf16 matrix cores (fp32 softmax).  R / A-bar stay fp32 always.
        self.forward_gemm_dtype = torch.float32
        self.attention_mma_bf16 = False
        # capture slabs follow ``set_body_dtype`` (image towers: the slabs are the resident bytes of a long sequence);
        # the 77-token text tower keeps fp32 slabs and its register-resident exact-fp32 attention kernels
        self._long_sequence_slabs_follow_body = attn_mask is None

    def set_body_dtype(self, dtype):
        """``torch.bfloat16``: everything the reference's half-precision body would run in half precision (all GEMMs, the
        attention products where the streaming kernels apply, the capture slabs); ``torch.float32``: the exact path.
        ``torch.float16``: the reference's OWN half-precision mode (``convert_weights``, CLIP/clip/model.py:381-402): every GEMM
        on the fp16 matrix cores with the weights rounded to fp16 exactly as ``convert_weights`` rounds them (fp32 accumulate;
        LayerNorm, softmax and the residual stream stay fp32 -- the reference computes LayerNorm in fp32 too, model.py:157-164),
        long-sequence capture slabs in fp16, and the relevancy chain with the reference's fp16 roundings (``half_chain``: R is
        created in the dtype of the probabilities, notebook cell 6:20,43).  Pinned on the reference's fp16 model in
        ``tests/test_gpu_clip.py`` (fixture clip_tiny_fp16.npz)."""
        if dtype not in (torch.float32, torch.bfloat16, torch.float16):
            raise ValueError("body dtype must be torch.float32, torch.bfloat16 or torch.float16")
        half = dtype == torch.bfloat16
        self.backward_gemm_dtype = self.forward_gemm_dtype = dtype
        self.attention_mma_bf16 = half
        self.half_chain = dtype == torch.float16
        if self._long_sequence_slabs_follow_body:
            self.capture_dtype = dtype
            self.buffers = None

    def _linear(self, x, lin_weight, lin_bias):
        return ops.linear(x, lin_weight, lin_bias, getattr(self, "forward_gemm_dtype", torch.float32))

    def _gemm(self, x, weight):
        """``x @ weight`` for the hand-written backward, in ``backward_gemm_dtype`` (cached converted weights)."""
        return ops.backward_gemm(x, weight, getattr(self, "backward_gemm_dtype", torch.float32))

    def _ensure_buffers(self, batch, n_tokens, device, shared_probs=False, grads=True):
        if self.buffers is None or not self.buffers.matches(self.layers, batch, self.heads, n_tokens, n_tokens, device,
                                                            shared_probs, self.capture_dtype, grads):
            self.buffers = CaptureBuffers(self.layers, batch, self.heads, n_tokens, n_tokens, device=device,
                                          shared_probs=shared_probs, dtype=self.capture_dtype, grads=grads)
        return self.buffers

    # ------------------------------------------------------------------------------------------------------------
    # Tape forward / hand-written backward -- the path ``interpret`` runs for BOTH towers.
    #
    # The explainability pass needs d(logit)/d(attention probabilities) only.  Autograd would rebuild a graph per call
    # (and the reference even runs one partial backward per layer); here the forward keeps a small tape and the backward
    # is the same chain of vector-Jacobian products written out: input-gradient GEMMs, the fused LayerNorm-backward +
    # residual kernel, the fused QuickGELU backward, our attention backward.  Weight gradients are never formed.
    #
    # Shared-forward mode (``x`` has batch 1, ``batch`` > 1): CLIP ``interpret`` repeats ONE image B times (notebook
    # cell 6:3); the B copies have identical activations and differ only in their upstream gradients.  The forward runs
    # ONCE, the backward runs at batch B with batch-stride-0 q / k / v / P and LayerNorm statistics shared by the batch.
    # ------------------------------------------------------------------------------------------------------------
    def _mask_for(self, blk, n_tokens, device):
        mask = blk.attn_mask
        if mask is None:
            return None
        if mask.device != device:
            blk.attn_mask = mask = mask.to(device=device, dtype=torch.float32)
        if mask.shape[-1] != n_tokens:
            mask = mask[:n_tokens, :n_tokens].contiguous()               # trimmed (padding-free) text batch
        return mask

    @torch.no_grad()
    def forward_tape(self, x, batch=None, first_grad_layer=0, grads=True, out_rows=None):
        """``x``: ``[Bx, N, E]`` block input (embedded, through ``ln_pre`` for the image tower).  ``batch``: how many
        upstream gradients ``backward_tape`` will carry (default ``Bx``; ``Bx == 1 < batch`` = shared-forward mode).
        Returns ``(y [Bx, N, E], tape)``; the probabilities of every block are in the capture slabs afterwards.
        ``grads=False``: no gradient slab is allocated (``backward_tape(..., rel_row=...)`` does not store dP).

        ``out_rows`` (``[Bx]`` long): the caller reads ONE token per sample from the tower output (class token / EOT token:
        CLIP/clip/model.py:235, 360).  Everything after the attention of the top block is row-wise, so ``out_proj``, ``ln_2``
        and the MLP of that block run on those Bx rows only (3 of its 4 GEMMs); ``y`` is then ``[Bx, E]`` -- those rows -- and
        ``backward_tape`` must be given the same rows as ``dy_rows``."""
        if not x.is_cuda:
            raise _lib.MMXError("the CLIP body runs its attention on the HIP capture op: move the model and "
                                "inputs to the MI355X (there is no CPU attention path)")
        Bx, N, E = x.shape
        batch = Bx if batch is None else batch
        shared = Bx == 1 and batch > 1
        if not shared and Bx != batch:
            raise ValueError("forward_tape: %d inputs for %d upstream gradients (only a single input can be shared)" % (Bx, batch))
        buffers = self._ensure_buffers(batch, N, x.device, shared_probs=shared, grads=grads)
        blocks = list(self.resblocks)
        mask = self._mask_for(blocks[0], N, x.device) if blocks else None
        tape = []
        # bf16 matrix cores for the attention products: the long-sequence streaming kernels only (the register-resident
        # head kernels of a short tower such as CLIP's 77-token text side stay exact fp32)
        mma = bool(getattr(self, "attention_mma_bf16", False)) and N > 128
        # every LayerNorm but the first is fused with the residual add that produces its input
        first = blocks[0].ln_1
        # bf16 body: what only feeds a GEMM (LayerNorm outputs, the MLP activation) leaves its kernel as bf16 -- no conversion passes
        hd = torch.bfloat16 if getattr(self, "forward_gemm_dtype", torch.float32) == torch.bfloat16 else torch.float32
        _, h1, mean1, rstd1 = ops.add_layernorm(x, None, first.weight, first.bias, first.eps, h_dtype=hd)
        for l, blk in enumerate(blocks):
            at = blk.attn
            qkv = self._linear(h1, at.in_proj_weight, at.in_proj_bias).view(Bx, N, 3, at.num_heads, at.head_dim)
            o = ops.attn_capture_fwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], buffers.probs[l], at.head_dim ** -0.5,
                                     _lib.SCALE_Q_FIRST, mask, layout="bnhd", mma_bf16=mma)
            if out_rows is not None and l + 1 == len(blocks):
                ar = torch.arange(Bx, device=x.device)
                x1, h2, mean2, rstd2 = ops.add_layernorm(
                    x[ar, out_rows], self._linear(o.view(Bx, N, E)[ar, out_rows], at.out_proj.weight, at.out_proj.bias),
                    blk.ln_2.weight, blk.ln_2.bias, blk.ln_2.eps, h_dtype=hd)
                m = self._linear(h2, blk.mlp.c_fc.weight, blk.mlp.c_fc.bias)
                mlp_out = self._linear(ops.quick_gelu_fwd(m, hd), blk.mlp.c_proj.weight, blk.mlp.c_proj.bias)
                # (x1, mean2, rstd2, m hold the Bx selected rows only: what _top_block_rows reads)
                tape.append((x, mean1, rstd1, qkv, x1, mean2, rstd2, m, o, out_rows) if l >= first_grad_layer else None)
                blk.attn_probs, blk.attn_grad = buffers.layer_probs(l), buffers.layer_grads(l)
                return x1 + mlp_out, tape
            x1, h2, mean2, rstd2 = ops.add_layernorm(x, self._linear(o.view(Bx, N, E), at.out_proj.weight, at.out_proj.bias),
                                                     blk.ln_2.weight, blk.ln_2.bias, blk.ln_2.eps, h_dtype=hd)
            m = self._linear(h2, blk.mlp.c_fc.weight, blk.mlp.c_fc.bias)
            mlp_out = self._linear(ops.quick_gelu_fwd(m, hd), blk.mlp.c_proj.weight, blk.mlp.c_proj.bias)
            tape.append((x, mean1, rstd1, qkv, x1, mean2, rstd2, m, o) if l >= first_grad_layer else None)
            blk.attn_probs, blk.attn_grad = buffers.layer_probs(l), buffers.layer_grads(l)
            if l + 1 < len(blocks):
                nxt = blocks[l + 1].ln_1
                x, h1, mean1, rstd1 = ops.add_layernorm(x1, mlp_out, nxt.weight, nxt.bias, nxt.eps, h_dtype=hd)
            else:
                x = x1 + mlp_out
        return x, tape

    def forward_shared(self, x, batch):
        """Shared-forward mode of ``forward_tape`` (kept under its round-1 name)."""
        if x.shape[0] != 1:
            raise ValueError("forward_shared takes the single shared sample")
        return self.forward_tape(x, batch)

    @torch.no_grad()
    def backward_tape(self, tape, dy, first_grad_layer=0, dy_rows=None, rel_row=None, dy_row_values=None):
        """``dy``: ``[B, N, E]`` upstream gradients w.r.t. the tower output; fills ``buffers.grads`` of every block
        ``>= first_grad_layer``.

        ``rel_row`` (``[B, N]`` fp32, bf16-body towers on the streaming kernels only): row-relevancy mode.  The caller
        wants ONE row of ``R = (I + A_top) ... (I + A_first)`` (CLIP ``interpret`` returns ``R[:, 0, 1:]``): that row is
        ``rel_row`` carried through the layers top-down -- the order this backward runs in -- as
        ``row <- row + row . mean_h clamp(dP * P, 0)``, reduced inside the attention backward kernel.  No gradient slab is
        written, no A-bar / R matrix is formed; the final row is returned.

        ``dy_rows`` (``[B]`` long, optional): promise that ``dy`` is zero outside row ``dy_rows[b]`` of sample ``b`` -- both
        CLIP towers read their feature from ONE token (class token / EOT token).  The top block's MLP and ``out_proj``
        vector-Jacobian products are row-wise, so they then run on those B rows instead of B*N (3 of the 4 GEMMs of
        that block); below the top block's attention the gradient is dense and everything runs in full.
        ``dy_row_values`` (``[B, E]``, with ``dy_rows``): those rows themselves -- ``dy`` may then be ``None`` (no dense zero
        tensor is built just to be gathered from again)."""
        if dy is None:
            if dy_rows is None or dy_row_values is None:
                raise ValueError("backward_tape: dy=None needs dy_rows and dy_row_values")
            if self.layers - 1 < first_grad_layer:         # nothing to explain (the tape holds None for every layer)
                return rel_row
            B, N, E = dy_row_values.shape[0], tape[-1][0].shape[1], dy_row_values.shape[1]
        else:
            B, N, E = dy.shape
        buffers = self.buffers
        top = self.layers - 1
        dx = dy
        top_rows = None                      # (rows, d_x1 of those rows): the top block's residual gradient, added after LN1'
        mma = bool(getattr(self, "attention_mma_bf16", False)) and N > 128
        # bf16 body on the streaming kernels: the gradients BETWEEN the GEMMs are bf16 (what the bf16 GEMMs produce and
        # consume; the elementwise kernels and the attention backward read / write bf16 directly -- no conversion
        # passes), the residual gradient stream (dx, d_x1) stays fp32
        # A short tower of a bf16 body (CLIP's 77-token text side: register-resident fp32 head kernels) runs the same bf16 stream: the
        # whole-head backward reads a bf16 d_o and writes bf16 dq | dk | dv around its exact-fp32 arithmetic (attention_head.hip, IOH;
        # round 4 -- before that two conversion passes per layer sat around it).
        stream16 = getattr(self, "backward_gemm_dtype", torch.float32) == torch.bfloat16
        head = self.resblocks[0].attn
        att16 = stream16 and (mma or (self.capture_dtype == torch.float32 and
                                      ops.head_kernel_shape(N, N, head.head_dim) and head.head_dim % 8 == 0))
        dx_h = None
        for l in range(top, first_grad_layer - 1, -1):
            blk = self.resblocks[l]
            at = blk.attn
            x, mean1, rstd1, qkv, x1, mean2, rstd2, m, o_fwd = tape[l][:9]
            shared = x.shape[0] != B
            if len(tape[l]) > 9 and (l != top or dy_rows is None):
                raise ValueError("backward_tape: the forward kept only the output rows of the top block (out_rows): "
                                 "pass the same rows as dy_rows")
            if l == top and dy_rows is not None:
                g = dy_row_values if dy_row_values is not None else dy[torch.arange(B, device=dy_rows.device), dy_rows]
                d_x1_r, d_o = self._top_block_rows(blk, tape[l], g, dy_rows, shared, N)
                d_x1, top_rows = None, (dy_rows, d_x1_r)
                if att16:
                    d_o = d_o.to(torch.bfloat16)
            elif stream16:
                if dx_h is None:
                    dx_h = dx.to(torch.bfloat16)
                d_a = ops.backward_gemm_bf16(dx_h, blk.mlp.c_proj.weight)
                d_m = ops.quick_gelu_bwd(m, d_a)
                d_h2 = ops.backward_gemm_bf16(d_m, blk.mlp.c_fc.weight)
                d_x1, d_x1_h = ops.layernorm_bwd_add_bf16(d_h2, x1, mean2, rstd2, blk.ln_2.weight, dx)
                d_o = ops.backward_gemm_bf16(d_x1_h, at.out_proj.weight)
            else:
                # MLP branch: x2 = x1 + c_proj(m * sigmoid(1.702 m)),  m = c_fc(ln_2(x1))
                d_a = self._gemm(dx, blk.mlp.c_proj.weight)
                d_m = ops.quick_gelu_bwd(m, d_a)          # shared mode: m [1, N, 4E] is broadcast inside the kernel
                d_h2 = self._gemm(d_m, blk.mlp.c_fc.weight)
                d_x1 = ops.layernorm_bwd_add(d_h2, x1, mean2, rstd2, blk.ln_2.weight, dx)   # dx + LN2'(d_h2), one pass
                # attention branch: x1 = x + out_proj(attn(ln_1(x)))
                d_o = self._gemm(d_x1, at.out_proj.weight)
            if stream16 and not att16 and d_o.dtype != torch.float32:
                d_o = d_o.float()
            d_o = d_o.view(B, N, at.num_heads, at.head_dim)
            need = l > first_grad_layer                                       # nothing below needs gradients
            dqkv = torch.empty(B, N, 3, at.num_heads, at.head_dim, dtype=d_o.dtype, device=d_o.device) if need else None
            out = (dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2]) if need else None
            res = ops.attn_capture_bwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], buffers.probs[l], d_o,
                                       buffers.grads[l] if buffers.grads is not None else None,
                                       at.head_dim ** -0.5, _lib.SCALE_Q_FIRST, need_dqkv=need, layout="bnhd", out=out,
                                       batch=B if shared else None, o=o_fwd, mma_bf16=mma, rel_row=rel_row)
            if rel_row is not None:
                rel_row = res[3]
            if not need:
                break
            if stream16:
                dq_in = dqkv.view(B, N, 3 * E)
                d_h1 = ops.backward_gemm_bf16(dq_in if att16 else dq_in.to(torch.bfloat16), at.in_proj_weight)
                dx, dx_h = ops.layernorm_bwd_add_bf16(d_h1, x, mean1, rstd1, blk.ln_1.weight, d_x1)
            else:
                d_h1 = self._gemm(dqkv.view(B, N, 3 * E), at.in_proj_weight)
                dx = ops.layernorm_bwd_add(d_h1, x, mean1, rstd1, blk.ln_1.weight, d_x1)
            if l == top and top_rows is not None:
                # the top block's residual gradient lives on one row per sample: added to LN1'(.) there (no dense zero tensor)
                ops.rows_add_(dx, *top_rows)
                if stream16:
                    dx_h = dx.to(torch.bfloat16)
        return rel_row

    def _top_block_rows(self, blk, entry, g, rows, shared, N):
        """MLP / ``out_proj`` backward of the top block on the one row per sample that carries a gradient (``g [B, E]``).
        Returns ``(d_x1 rows [B, E], d_o dense [B, N, E])`` -- the attention backward wants every row of ``d_o``, zero elsewhere."""
        x, mean1, rstd1, qkv, x1, mean2, rstd2, m, o_fwd = entry[:9]
        B = g.shape[0]
        ar = torch.arange(B, device=g.device)
        src = torch.zeros_like(rows) if shared else ar                     # sample index into the (shared) tape
        if len(entry) > 9:                                                   # forward_tape(out_rows=...): rows only on the tape
            m_r, x1_r, mean2_r, rstd2_r = m[src], x1[src], mean2.reshape(-1)[src], rstd2.reshape(-1)[src]
        else:
            flat = src * N + rows                                            # row of the [Bx*N] statistics
            m_r, x1_r, mean2_r, rstd2_r = m[src, rows], x1[src, rows], mean2.reshape(-1)[flat], rstd2.reshape(-1)[flat]
        d_a = self._gemm(g, blk.mlp.c_proj.weight)
        d_m = ops.quick_gelu_bwd(m_r, d_a) if d_a.dtype == torch.float32 and m_r.numel() % 4 == 0 else \
            d_a * (torch.sigmoid(1.702 * m_r) * (1 + 1.702 * m_r * (1 - torch.sigmoid(1.702 * m_r))))
        d_h2 = self._gemm(d_m, blk.mlp.c_fc.weight)
        d_x1_r = ops.layernorm_bwd_add(d_h2, x1_r, mean2_r, rstd2_r, blk.ln_2.weight, g)
        return d_x1_r, ops.rows_to_dense(self._gemm(d_x1_r, blk.attn.out_proj.weight), rows, N)

    def backward_shared(self, tape, dy, first_grad_layer=0):
        """Round-1 name of ``backward_tape``."""
        return self.backward_tape(tape, dy, first_grad_layer)

    def forward(self, x, capture_only=False, first_grad_layer=0):
        """``x``: ``[B, N, E]``.  ``capture_only``: only d(loss)/d(probs) of blocks ``>= first_grad_layer`` is wanted
        (``start_layer`` of the explainability pass): blocks below run without an autograd graph, the graph is cut at
        the input of block ``first_grad_layer`` and that block skips its dq/dk/dv."""
        if not x.is_cuda:
            raise _lib.MMXError("the CLIP body runs its attention on the HIP capture op: move the model and "
                                "inputs to the MI355X (there is no CPU attention path)")
        buffers = self._ensure_buffers(x.shape[0], x.shape[1], x.device)
        if not capture_only:
            for l, blk in enumerate(self.resblocks):
                x = blk(x, buffers, l)
            return x
        with torch.no_grad():
            for l in range(first_grad_layer):
                x = self.resblocks[l](x, buffers, l)
        x = x.detach().requires_grad_(True)
        for l in range(first_grad_layer, self.layers):
            x = self.resblocks[l](x, buffers, l, need_dqkv=(l != first_grad_layer))
        return x


class VisualTransformer(nn.Module):
    """ViT image tower (CLIP/clip/model.py:211-246)."""

    def __init__(self, input_resolution, patch_size, width, layers, heads, output_dim):
        super().__init__()
        self.input_resolution, self.output_dim = input_resolution, output_dim
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))

    def _patchify(self, x):
        """``conv1`` has kernel == stride == patch, i.e. it is a GEMM over flattened patches.  Doing it as one keeps
        MIOpen (solver search, and its naive fp32 fallback kernel: 24 ms per call here) off the path."""
        B, C, Hh, Ww = x.shape
        p = self.conv1.kernel_size[0]
        gh, gw = Hh // p, Ww // p
        patches = x.reshape(B, C, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * p * p)
        return F.linear(patches, self.conv1.weight.reshape(self.conv1.out_channels, -1))

    def _embed(self, x):
        x = self._patchify(x)                                               # [B, grid^2, width]
        cls = self.class_embedding.to(x.dtype).expand(x.shape[0], 1, -1)
        return self.ln_pre(torch.cat([cls, x], dim=1) + self.positional_embedding.to(x.dtype))

    def forward(self, x, capture_only=False, first_grad_layer=0):
        x = self.transformer(self._embed(x), capture_only=capture_only, first_grad_layer=first_grad_layer)
        x = self.ln_post(x[:, 0, :])
        return x @ self.proj if self.proj is not None else x

    def row_relevancy_ok(self):
        """Can ``backward_tape(..., cls_row=True)`` run?  (bf16-body tower on the long-sequence streaming kernels.)"""
        t = self.transformer
        n_tokens = self.positional_embedding.shape[0]
        return bool(getattr(t, "attention_mma_bf16", False)) and n_tokens > 128

    @torch.no_grad()
    def forward_tape(self, image, batch=None, first_grad_layer=0, grads=True):
        """``image [Bx, 3, R, R]`` -> ``(features [Bx, output_dim], state)``; ``Bx == 1 < batch``: shared-forward mode
        (see ``Transformer.forward_tape``)."""
        x = self._embed(image)
        y, tape = self.transformer.forward_tape(x, batch, first_grad_layer, grads=grads,
                                                out_rows=torch.zeros(x.shape[0], dtype=torch.long, device=x.device))
        cls = y                                                              # [Bx, width]: the class-token rows
        _, f, mean, rstd = ops.add_layernorm(cls, None, self.ln_post.weight, self.ln_post.bias, self.ln_post.eps)
        return f @ self.proj, (tape, x.shape, cls, mean, rstd)

    def forward_shared(self, image, batch):
        return self.forward_tape(image, batch)

    @torch.no_grad()
    def backward_tape(self, state, d_features, first_grad_layer=0, cls_row=False):
        """``d_features [B, output_dim]``: per-sample upstream gradients of the image features.
        ``cls_row=True``: row-relevancy mode (``Transformer.backward_tape``): returns row 0 (the class token's) of the
        tower's relevancy matrix, ``[B, N]``, instead of filling gradient slabs."""
        tape, y_shape, cls, mean, rstd = state
        B = d_features.shape[0]
        d_f = torch.matmul(d_features, self.proj.t())
        # LayerNorm' against the (shared) class-token rows' statistics; only the class token feeds the features
        d_cls = ops.layernorm_bwd_add(d_f, cls, mean, rstd, self.ln_post.weight)
        rel_row = None
        if cls_row:
            rel_row = torch.zeros(B, y_shape[1], dtype=torch.float32, device=d_f.device)
            rel_row[:, 0] = 1.0                                              # e_0: row 0 of the identity R starts from
        return self.transformer.backward_tape(tape, None, first_grad_layer,
                                              dy_rows=torch.zeros(B, dtype=torch.long, device=d_f.device), rel_row=rel_row,
                                              dy_row_values=d_cls)

    def backward_shared(self, state, d_features, first_grad_layer=0):
        return self.backward_tape(state, d_features, first_grad_layer)


class CLIP(nn.Module):
    """Same constructor signature as the reference ``CLIP`` (CLIP/clip/model.py:249-262); ViT towers only."""

    def __init__(self, embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size,
                 context_length, vocab_size, transformer_width, transformer_heads, transformer_layers):
        super().__init__()
        if isinstance(vision_layers, (tuple, list)):
            raise NotImplementedError("ModifiedResNet towers have no attention stack; use a ViT config")
        self.context_length = context_length
        self.visual = VisualTransformer(image_resolution, vision_patch_size, vision_width, vision_layers,
                                        vision_width // 64, embed_dim)
        self.transformer = Transformer(transformer_width, transformer_layers, transformer_heads,
                                       attn_mask=self.build_attention_mask())
        self.vocab_size = vocab_size
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, transformer_width))
        self.ln_final = nn.LayerNorm(transformer_width)
        self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        self.capture_only = False
        self.first_grad_layers = (0, 0)     # (image tower, text tower) start layers of a capture-only pass
        self.initialize_parameters()

    def initialize_parameters(self):
        """Same distributions as the reference (model.py:305-332)."""
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        w, n = self.transformer.width, self.transformer.layers
        proj_std, attn_std, fc_std = (w ** -0.5) * ((2 * n) ** -0.5), w ** -0.5, (2 * w) ** -0.5
        for blk in self.transformer.resblocks:
            nn.init.normal_(blk.attn.in_proj_weight, std=attn_std)
            nn.init.normal_(blk.attn.out_proj.weight, std=proj_std)
            nn.init.normal_(blk.mlp.c_fc.weight, std=fc_std)
            nn.init.normal_(blk.mlp.c_proj.weight, std=proj_std)
        nn.init.normal_(self.text_projection, std=w ** -0.5)

    def set_body_dtype(self, dtype):
        """``torch.bfloat16``: BASELINE config 5's bf16 body -- both towers' GEMMs on the bf16 matrix cores (fp32
        accumulate), the image tower's attention products too, its capture slabs in bf16; relevancy (A-bar, R) stays fp32.
        ``torch.float16``: the reference's own half-precision mode (``convert_weights``, CLIP/clip/model.py:381-402): fp16 GEMMs
        and the fp16 relevancy chain of notebook cell 6:20,43 -- ``interpret`` then returns fp16 like the reference does.
        ``torch.float32``: back to the exact path."""
        self.visual.transformer.set_body_dtype(dtype)
        self.transformer.set_body_dtype(dtype)

    def build_attention_mask(self):
        """Additive causal mask, -inf above the diagonal (model.py:334-340)."""
        return torch.full((self.context_length, self.context_length), float("-inf")).triu_(1)

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    def encode_image(self, image):
        return self.visual(image.type(self.dtype), capture_only=self.capture_only,
                           first_grad_layer=self.first_grad_layers[0])

    def encode_text(self, text, n_tokens=None):
        """``n_tokens``: run only the first ``n_tokens`` positions (must cover every sequence's EOT token).  With the
        causal mask nothing after a sequence's EOT can influence its feature, so this is exact, not an approximation."""
        if n_tokens is not None and n_tokens < text.shape[1]:
            text = text[:, :n_tokens]
        n = text.shape[1]
        x = self.token_embedding(text).type(self.dtype) + self.positional_embedding[:n].type(self.dtype)
        x = self.transformer(x, capture_only=self.capture_only, first_grad_layer=self.first_grad_layers[1])
        x = self.ln_final(x)
        # features at the EOT token = highest token id in each sequence (model.py:360)
        return x[torch.arange(x.shape[0], device=x.device), text.argmax(dim=-1)] @ self.text_projection

    @torch.no_grad()
    def encode_text_tape(self, text, n_tokens=None, first_grad_layer=0):
        """``encode_text`` on the tape path -> ``(features [B, embed_dim], state)`` for ``backward_text_tape``."""
        if n_tokens is not None and n_tokens < text.shape[1]:
            text = text[:, :n_tokens]
        n = text.shape[1]
        x = self.token_embedding(text).type(self.dtype) + self.positional_embedding[:n].type(self.dtype)
        eot = text.argmax(dim=-1)                                            # model.py:360
        rows, tape = self.transformer.forward_tape(x, first_grad_layer=first_grad_layer, out_rows=eot)
        _, f, mean, rstd = ops.add_layernorm(rows, None, self.ln_final.weight, self.ln_final.bias, self.ln_final.eps)
        return f @ self.text_projection, (tape, x.shape, rows, mean, rstd, eot)

    @torch.no_grad()
    def backward_text_tape(self, state, d_features, first_grad_layer=0):
        """``d_features [B, embed_dim]`` -> fills the text tower's gradient slabs (``ln_final`` is row-wise, so only the
        EOT rows carry a gradient into the stack)."""
        tape, y_shape, rows, mean, rstd, eot = state
        d_f = torch.matmul(d_features, self.text_projection.t())
        d_rows = ops.layernorm_bwd_add(d_f, rows, mean, rstd, self.ln_final.weight)
        self.transformer.backward_tape(tape, None, first_grad_layer, dy_rows=eot, dy_row_values=d_rows)

    def forward(self, image, text):
        return self.logits(self.encode_image(image), self.encode_text(text))

    def logits(self, image_features, text_features):
        """Cosine-similarity logits (model.py:369-378) from un-normalised features."""
        image_features = image_features / image_features.norm(dim=-1, keepdim=True)
        text_features = text_features / text_features.norm(dim=-1, keepdim=True)
        logit_scale = self.logit_scale.exp()
        logits_per_image = logit_scale * image_features @ text_features.t()
        logits_per_text = logit_scale * text_features @ image_features.t()
        return logits_per_image, logits_per_text


def build_model(state_dict):
    """Infer the ViT config from a state dict like the reference's ``build_model`` (model.py:405-442), fp32."""
    if "visual.proj" not in state_dict:
        raise NotImplementedError("only ViT CLIP checkpoints carry an attention stack to explain")
    vision_width = state_dict["visual.conv1.weight"].shape[0]
    vision_layers = len([k for k in state_dict if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    vision_patch_size = state_dict["visual.conv1.weight"].shape[-1]
    grid_size = round((state_dict["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    embed_dim = state_dict["text_projection"].shape[1]
    context_length = state_dict["positional_embedding"].shape[0]
    vocab_size = state_dict["token_embedding.weight"].shape[0]
    transformer_width = state_dict["ln_final.weight"].shape[0]
    transformer_layers = len({k.split(".")[2] for k in state_dict if k.startswith("transformer.resblocks")})
    model = CLIP(embed_dim, vision_patch_size * grid_size, vision_layers, vision_width, vision_patch_size,
                 context_length, vocab_size, transformer_width, transformer_width // 64, transformer_layers)
    sd = {k: v.float() for k, v in state_dict.items() if k not in ("input_resolution", "context_length", "vocab_size")}
    model.load_state_dict(sd)
    return model.eval()


CONFIGS = {
    # (embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size,
    #  context_length, vocab_size, transformer_width, transformer_heads, transformer_layers)
    "ViT-B/32": (512, 224, 12, 768, 32, 77, 49408, 512, 8, 12),
    "ViT-B/16": (512, 224, 12, 768, 16, 77, 49408, 512, 8, 12),
    "ViT-L/14@336": (768, 336, 24, 1024, 14, 77, 49408, 768, 12, 12),
}


def random_init(name="ViT-B/32", seed=0):
    """Random-init model of a named architecture (no pretrained weights are available offline)."""
    torch.manual_seed(seed)
    return CLIP(*CONFIGS[name]).float().eval()
