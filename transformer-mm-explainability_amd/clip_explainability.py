"""Drop-in ``interpret`` for CLIP -- same signatures as the reference, relevancy chain on the fused HIP kernel.

Reference entry points mirrored here:
  * ``interpret(image, texts, model, device, start_layer=-1, start_layer_text=-1)``
    -- CLIP_explainability.ipynb cell 6 (batched: one image, B texts) -> ``(text_relevance [B,Nt,Nt], image_relevance [B,Ni-1])``
  * ``interpret_single(image, text, model, device, index=None)``
    -- CLIP/example.py:8-32 (one image, K texts, explains ``logits_per_image[0, index]``) -> ``image_relevance [Ni-1]``
  * ``text_scores(text_encoding, R_text)`` -- notebook cell 8:5-7 post-processing (on device)

What is different under the hood (results agree to fp32 rounding, see tests/test_gpu_clip.py):
  * the reference issues one ``torch.autograd.grad`` per layer (24 partial backward passes); here ONE hand-written
    backward per tower (``clip_model.Transformer.forward_tape`` / ``backward_tape``: no autograd graph through the bodies,
    no weight gradients, fused LayerNorm / QuickGELU / attention-capture kernels) fills every layer's gradient slab;
  * the per-layer reshape/mul/clamp/mean/bmm/add launches are one ``relevancy_self_chain`` launch per tower.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .rules import frozen_parameters as _Frozen

def _plan(buffers, first, last, batch_size, shared, half_chain=False, causal=False):
    """Prepared chain launch, cached on the slab object (the slabs keep their addresses from call to call).  ``half_chain``:
    the reference's fp16 chain (a model after ``set_body_dtype(torch.float16)``; ``ops.ChainPlan``).  ``causal``: the tower's
    attention carries CLIP's causal mask (the text tower, CLIP/clip/model.py:334-340): its probabilities are exact zeros above the
    diagonal and the chain kernel does not read that half of either slab (``MMX_CHAIN_CAUSAL``)."""
    cache = buffers.__dict__.setdefault("_chain_plans", {})
    key = (first, last, batch_size, shared, bool(half_chain), bool(causal))
    if key not in cache:
        cache[key] = ops.ChainPlan([buffers.probs[l] for l in range(first, last)],
                                   [buffers.grads[l] for l in range(first, last)], batch_size, shared_attn=shared,
                                   half_chain=half_chain, causal=causal)
    return cache[key]


def _is_causal_tower(tower):
    """True iff every block of ``tower`` runs under an additive mask that is -inf strictly above the diagonal (CLIP's
    ``build_attention_mask``, CLIP/clip/model.py:334-340): its probabilities are then exact zeros there and the chain may skip that
    half (``_plan(causal=True)``).  Checked on the mask tensors themselves, once per mask object (one device -> host read, made during
    the warm-up call that precedes any hipGraph capture); anything else -- no mask, another mask -- is not causal."""
    blocks = list(tower.resblocks)
    masks = [getattr(blk, "attn_mask", None) for blk in blocks]
    if not masks or any(m is None for m in masks):
        return False
    key = tuple(id(m) for m in masks)
    cached = tower.__dict__.get("_causal_mask_check")
    if cached is None or cached[0] != key:
        ok = True
        for m in {id(m): m for m in masks}.values():
            n = m.shape[-1]
            upper = torch.ones(n, n, dtype=torch.bool, device=m.device).triu_(1)
            ok = ok and m.dim() == 2 and m.shape[0] == n and bool(torch.isneginf(m[upper]).all())
        tower.__dict__["_causal_mask_check"] = cached = (key, ok)
    return cached[1]


def _side_stream(device):
    """The side stream of the image tower (``ops.side_stream``: shared by eager calls, private to a capture inside one)."""
    return ops.side_stream(device)


def interpret(image, texts, model, device, start_layer=-1, start_layer_text=-1, share_image_forward=True,
              trim_text_padding=False, _n_text=None, overlap_towers=True, image_chain_on_main=False):
    """CLIP_explainability.ipynb cell 6.  ``image``: ``[1,3,R,R]``, ``texts``: ``[B, context]`` token ids.

    ``share_image_forward`` (extra keyword, default on): the reference repeats the ONE image B times (cell 6:3) and
    runs the image tower on B identical copies.  Here its forward runs once and only the backward -- whose upstream
    gradients do differ per text -- runs at batch B (``clip_model.Transformer.forward_shared``); results are the
    same to fp32 rounding (``tests/test_gpu_clip.py``).  ``False`` runs the B copies like the reference.

    ``trim_text_padding`` (extra keyword, default OFF): CLIP pads every caption to 77 tokens and the reference runs the
    text tower over all of them.  The mask is causal and the feature is read at the EOT token, so positions after the
    last EOT of the batch influence nothing: their gradient rows are exactly zero and the returned ``R_text`` is the
    identity there.  With the flag the text tower runs only on the first ``max(EOT)+1`` positions and the ``[B,77,77]``
    result is assembled around that block -- identical output, 5-6x less text-tower work for caption-length inputs.

    ``overlap_towers`` (extra keyword, default on): image tower on a side stream, text tower on the current one (they
    only meet in the similarity head); ``False`` runs them back to back on the current stream.  Same results bit for bit.

    ``image_chain_on_main`` (extra keyword, default off): launch the image tower's chain kernel on the CURRENT stream after
    the text tower's work instead of on the side stream next to the text tower's GEMMs (where it measured 118 us instead of
    its stand-alone 44 us); ``bench.py`` reports the step rate of both orders.  Same results.
    """
    batch_size = texts.shape[0]
    sl = model.visual.transformer.layers - 1 if start_layer == -1 else start_layer
    slt = model.transformer.layers - 1 if start_layer_text == -1 else start_layer_text
    n_text = _n_text
    if trim_text_padding and n_text is None:
        n_text = int(texts.argmax(dim=-1).max()) + 1                                      # one D2H read of the ids
    shared = share_image_forward and image.shape[0] == 1 and batch_size > 1
    # Both towers on the tape path (clip_model.Transformer.forward_tape / backward_tape): no autograd graph through the
    # bodies, no weight gradients; only the cosine-similarity head (B x embed_dim features) goes through autograd.
    images = image.type(model.dtype) if shared or image.shape[0] == batch_size else \
        image.type(model.dtype).repeat(batch_size, 1, 1, 1)                                # cell 6:3
    # The two towers are independent up to the similarity head.  The image tower runs on a side stream: its shared
    # forward is ~120 batch-1 launches of a few microseconds (latency-bound) that hide completely behind the text
    # tower's GEMMs, and in the backward the kernel-boundary gaps of one tower are filled by the other's kernels.
    # Fork / join with stream waits only, so the whole thing still captures into one hipGraph (parallel branches).
    main = torch.cuda.current_stream()
    side = _side_stream(texts.device) if overlap_towers else main
    side.wait_stream(main)
    # bf16-body image tower (cfg 5): only R[:, 0, 1:] is returned, so the relevancy ROW is carried through the backward
    # (clip_model.Transformer.backward_tape, rel_row) -- no gradient slabs, no A-bar / R matrices for the image side
    row_mode = shared and model.visual.row_relevancy_ok()
    with torch.cuda.stream(side):
        img_feat, img_state = model.visual.forward_tape(images, batch_size, sl, grads=not row_mode)
    txt_feat, txt_state = model.encode_text_tape(texts, n_text, slt)
    main.wait_stream(side)
    with torch.enable_grad():
        image_features = img_feat.detach().expand(batch_size, -1).contiguous().requires_grad_(True)   # per-sample leaves
        text_features = txt_feat.detach().requires_grad_(True)
        logits_per_image, _ = model.logits(image_features, text_features)
        # one_hot = sum_i logits_per_image[i, i]  (cell 6:6-10)  ->  d one_hot / d logits = I
        eye = torch.eye(batch_size, dtype=torch.float32, device=texts.device)
        torch.autograd.backward(logits_per_image, grad_tensors=eye, inputs=[image_features, text_features])
    side.wait_stream(main)
    with torch.cuda.stream(side):
        vis = model.visual.transformer
        if row_mode:
            image_relevance = model.visual.backward_tape(img_state, image_features.grad, sl, cls_row=True)[:, 1:]
        else:
            model.visual.backward_tape(img_state, image_features.grad, sl)
            if not image_chain_on_main:
                R = _plan(vis.buffers, sl, vis.layers, batch_size, vis.buffers.shared_probs and batch_size > 1,
                          getattr(vis, "half_chain", False)).launch()
                image_relevance = R[:, 0, 1:]
    model.backward_text_tape(txt_state, text_features.grad, slt)
    txt = model.transformer
    R_text = _plan(txt.buffers, slt, txt.layers, batch_size, False, getattr(txt, "half_chain", False),
                   causal=_is_causal_tower(txt)).launch()
    main.wait_stream(side)
    if image_chain_on_main and not row_mode:
        R = _plan(vis.buffers, sl, vis.layers, batch_size, vis.buffers.shared_probs and batch_size > 1,
                  getattr(vis, "half_chain", False)).launch()
        image_relevance = R[:, 0, 1:]
    if R_text.shape[-1] != texts.shape[1]:       # trimmed run: the rest of the [B, 77, 77] matrix is the identity
        n = R_text.shape[-1]
        full = torch.eye(texts.shape[1], dtype=R_text.dtype, device=R_text.device).repeat(batch_size, 1, 1)
        full[:, :n, :n] = R_text
        R_text = full
    return R_text, image_relevance


class GraphedInterpret:
    """``interpret`` captured once into a hipGraph and replayed: the step is ~600 short launches (PyTorch body ops, our
    attention / chain kernels) and an eager step is bound by the Python + launch path on the host (~16 ms) rather than by
    the GPU; a replay costs one ``hipGraphLaunch``.  Shapes, ``start_layer``s and flags are fixed at construction;
    new ``image`` / ``texts`` values are copied into the captured input buffers.

        run = GraphedInterpret(model, image, texts, start_layer=0, start_layer_text=0)
        R_text, image_relevance = run(image2, texts2)      # same results as interpret(image2, texts2, model, device, 0, 0)

    The returned tensors are the graph's output buffers (overwritten by the next call; ``.clone()`` to keep them).
    """

    def __init__(self, model, image, texts, start_layer=-1, start_layer_text=-1, share_image_forward=True,
                 trim_text_padding=False, warmup=3, image_chain_on_main=False):
        self.model = model
        self.image = image.clone()
        self.texts = texts.clone()
        kw = dict(start_layer=start_layer, start_layer_text=start_layer_text, share_image_forward=share_image_forward,
                  image_chain_on_main=image_chain_on_main)
        # a trimmed run fixes the number of text positions at capture time: it must cover every later caption
        self.n_text = int(texts.argmax(dim=-1).max()) + 1 if trim_text_padding else None
        self._call = lambda: interpret(self.image, self.texts, model, self.image.device, _n_text=self.n_text, **kw)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._call()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: a collective backend's watchdog thread (RCCL, one rank per GPU) may poll events while this
        # thread captures; only calls made by the capturing thread must be capture-safe
        with ops.graph_capture(self.graph):
            self.outputs = self._call()
        # The graph bakes in the raw addresses of the towers' capture slabs and of the cached chain plans' scratch.  Pin
        # them: a later interpret() / GraphedInterpret on the same model with another batch, sharing mode or text length
        # makes ``Transformer._ensure_buffers`` install NEW slabs, and without this reference the old ones would be
        # freed under the graph.  ``__call__`` re-installs the pinned slabs so that ``blk.attn_probs`` / ``attn_grad``
        # and a following eager call see what the replay wrote.
        vis, txt = model.visual.transformer, model.transformer
        self._pinned = (vis.buffers, txt.buffers)
        self._pinned_scratch = ops.pinned_state()          # grow-only scratch buffers the graph has raw addresses of

    def _reinstall(self):
        for tr, buf in zip((self.model.visual.transformer, self.model.transformer), self._pinned):
            if tr.buffers is not buf:
                tr.buffers = buf
                for l, blk in enumerate(tr.resblocks):
                    blk.attn_probs, blk.attn_grad = buf.layer_probs(l), buf.layer_grads(l)

    def __call__(self, image=None, texts=None):
        self._reinstall()
        if image is not None:
            self.image.copy_(image)
        if texts is not None:
            if self.n_text is not None and int(texts.argmax(dim=-1).max()) + 1 > self.n_text:
                raise ValueError("caption longer than the %d positions this graph was captured for" % self.n_text)
            self.texts.copy_(texts)
        self.graph.replay()
        return self.outputs


def interpret_single(image, text, model, device, index=None):
    """CLIP/example.py:8-32 without the plotting: returns ``image_relevance [Ni-1]`` (``R[0,0]`` zeroed first)."""
    prev = (model.capture_only, model.first_grad_layers)
    model.capture_only, model.first_grad_layers = True, (0, model.transformer.layers)   # text tower: no gradient work
    try:
        with _Frozen(model), torch.enable_grad():
            logits_per_image, _ = model(image, text)
            if index is None:
                index = int(np.argmax(logits_per_image.detach().cpu().numpy(), axis=-1).reshape(-1)[0])
            one_hot = torch.zeros_like(logits_per_image)
            one_hot[0, index] = 1
            torch.autograd.backward(logits_per_image, grad_tensors=one_hot)
    finally:
        model.capture_only, model.first_grad_layers = prev
    vis = model.visual.transformer
    vb = vis.buffers
    # example.py flattens batch*heads into one head axis (cam.reshape(-1, N, N).mean(0)): batch_size = 1 here
    R = ops.relevancy_self_chain([vb.probs[l] for l in range(vis.layers)],
                                 [vb.grads[l] for l in range(vis.layers)], 1)[0]
    R[0, 0] = 0
    return R[0, 1:]


def text_scores(text_encoding, R_text):
    """Notebook cell 8:5-7 for one sample: row of the EOT token, columns ``1:EOT``, normalised by their sum."""
    cls_idx = int(text_encoding.argmax(dim=-1))
    row = R_text[cls_idx, 1:cls_idx]
    return (row / row.sum()).flatten()


def image_heatmap(image_relevance, size=224):
    """Notebook cell 7:14-18: reshape to the patch grid, bilinear upsample, min-max normalise -> ``[size, size]``
    (``postprocess.image_heatmaps`` does a whole batch in one launch)."""
    from .postprocess import image_heatmaps
    return image_heatmaps(image_relevance.reshape(-1), size)
