"""DETR relevancy generators -- the reference's ``DETR/modules/ExplanationGenerator.py`` surface on the HIP kernels.

Same class / method names, arguments, defaults, return shapes and attribute side effects (``self.R_i_i``,
``self.R_q_q``, ``self.R_q_i``) as the reference.  ``model`` is duck-typed exactly as there:
``model(img) -> {'pred_logits': [1, Q, C+1]}``, ``model.transformer.encoder.layers[i].self_attn``,
``model.transformer.decoder.layers[i].{self_attn, multihead_attn}``, each with ``get_attn()`` /
``get_attn_gradients()`` returning ``[B*H, Nq, Nk]`` device tensors (``attention_modules.MultiheadAttention`` provides
them straight from the capture slabs).

What changes underneath: the 6-layer encoder chain is ONE ``relevancy_self_chain`` call, each decoder self-attention
block is one call carrying both right-hand sides (rules 6+7), rule 10 runs on the MFMA matmul + row-normalise kernels.
LRP variants (``use_lrp=True`` -- the reference's DEFAULT --, ``generate_transformer_att``, ``generate_partial_lrp``): the
rule kernels take any cam, so these run exactly the reference's schedule on ``get_attn_cam()`` instead of ``get_attn()`` after
the body's LRP pass (``model.relprop(one_hot, alpha=1, target_index=..., target_class=...)`` filling ``save_attn_cam`` on
every attention module, as the reference's LRP layer library does, DETR/modules/layers.py:770-801).  ``detr_model``'s body
has that pass (``DETRFromFeatures.relprop``: closed-form layer rules + HIP attention-core kernels, ``lrp.py`` /
``csrc/attention_lrp.hip``), so ``Generator(detr_model).generate_ours(img, t)`` runs with its default arguments; a body
without ``relprop`` makes these entry points raise ``NotImplementedError`` naming the missing method, before any work.
"""
from __future__ import annotations

import collections

import torch

from . import ops, rules
from .rules import avg_heads, compute_rollout_attention, handle_residual  # noqa: F401  (re-exported like the reference module)

apply_self_attention_rules = rules.apply_self_attention_rules
apply_mm_attention_rules = rules.apply_mm_attention_rules_detr


def _logits_for_backward(model, img):
    """Forward pass the one-hot backward will run through; no rule reads a weight gradient, so parameters are frozen
    while the graph is built whenever the body allows it (``rules.forward_for_backward``)."""
    return rules.forward_for_backward(model, lambda: model(img)["pred_logits"])


def _one_hot_backward(model, outputs, target_index, index, backward=True):
    """DETR/modules/ExplanationGenerator.py:153-163: one-hot on (query, class), zero_grad, backward.
    Returns ``(index, one_hot_vector)``."""
    if index is None:
        index = outputs[0, target_index, :-1].max(1)[1]
    one_hot = torch.zeros_like(outputs)
    one_hot[0, target_index, index] = 1
    if backward:
        loss = torch.sum(one_hot * outputs)
        model.zero_grad()
        loss.backward(retain_graph=True)
    return index, one_hot


def _require_relprop(model):
    if not hasattr(model, "relprop"):
        raise NotImplementedError(
            "use_lrp=True / transformer_att / partial_lrp read LRP attention cams (get_attn_cam) that the body's "
            "relprop() must produce; %s has no relprop().  Use use_lrp=False (the evaluators' 'ours_no_lrp'), or plug "
            "a body built on an LRP layer library (reference: DETR/modules/layers.py)." % type(model).__name__)


def _relprop(model, one_hot_vector, target_index, index):
    """The reference's LRP pass (DETR/modules/ExplanationGenerator.py:165-166, 84, 213): the BODY's ``relprop`` leaves an
    ``attn_cam`` on every attention module (``save_attn_cam``, DETR/modules/layers.py:776)."""
    _require_relprop(model)
    model.relprop(one_hot_vector.clone().detach(), alpha=1, target_index=target_index, target_class=index)


def _cam(module, use_lrp):
    """``get_attn_cam()`` on the LRP route, ``get_attn()`` otherwise (reference :113-116, 122-125, 132-135)."""
    return (module.get_attn_cam() if use_lrp else module.get_attn()).detach()


class Generator:
    def __init__(self, model):
        self.model = model
        self.model.eval()

    def forward(self, input_ids, attention_mask):
        return self.model(input_ids, attention_mask)

    # ------------------------------------------------------------------ ours (rules 5, 6, 7, 10)
    def handle_self_attention_image(self, blocks):
        """Encoder chain (reference :110-118) in one launch."""
        attn = [_cam(blk.self_attn, self.use_lrp) for blk in blocks]
        grad = [blk.self_attn.get_attn_gradients().detach() for blk in blocks]
        self.R_i_i = ops.relevancy_self_chain(attn, grad, 1, R_init=self.R_i_i)[0]

    def handle_co_attn_self_query(self, block):
        """Rules 6+7 (reference :120-129): ``R_q_q += cam@R_q_q; R_q_i += cam@R_q_i``."""
        attn = _cam(block.self_attn, self.use_lrp)
        grad = block.self_attn.get_attn_gradients().detach()
        R_qq, R_qi = ops.relevancy_self_chain([attn], [grad], 1, R_init=self.R_q_q, R_sq_init=self.R_q_i)
        self.R_q_q, self.R_q_i = R_qq[0], R_qi[0]

    def handle_co_attn_query(self, block):
        """Rule 10 (reference :131-140)."""
        cam_q_i = avg_heads(_cam(block.multihead_attn, self.use_lrp), block.multihead_attn.get_attn_gradients().detach())
        self.R_q_i = self.R_q_i + apply_mm_attention_rules(self.R_q_q, self.R_i_i, cam_q_i,
                                                           apply_normalization=self.normalize_self_attention,
                                                           apply_self_in_rule_10=self.apply_self_in_rule_10)

    def generate_ours(self, img, target_index, index=None, use_lrp=True, normalize_self_attention=True,
                      apply_self_in_rule_10=True):
        if use_lrp:
            _require_relprop(self.model)                 # fail before the forward / backward, not after
        self.use_lrp = use_lrp
        self.normalize_self_attention = normalize_self_attention
        self.apply_self_in_rule_10 = apply_self_in_rule_10
        outputs = _logits_for_backward(self.model, img)
        index, one_hot_vector = _one_hot_backward(self.model, outputs, target_index, index)
        if use_lrp:
            _relprop(self.model, one_hot_vector, target_index, index)

        decoder_blocks = self.model.transformer.decoder.layers
        encoder_blocks = self.model.transformer.encoder.layers
        ref = encoder_blocks[0].self_attn.get_attn()
        image_bboxes = ref.shape[-1]
        queries_num = decoder_blocks[0].self_attn.get_attn().shape[-1]
        self.R_i_i = torch.eye(image_bboxes, image_bboxes, device=ref.device)
        self.R_q_q = torch.eye(queries_num, queries_num, device=ref.device)
        self.R_q_i = torch.zeros(queries_num, image_bboxes, device=ref.device)

        self.handle_self_attention_image(encoder_blocks)
        for blk in decoder_blocks:
            self.handle_co_attn_self_query(blk)
            self.handle_co_attn_query(blk)
        aggregated = self.R_q_i.unsqueeze_(0)
        return aggregated[:, target_index, :].unsqueeze_(0).detach()

    overlap_rules = True    # rows_only + shared forward: rule kernels start on a side stream as soon as their slabs are complete

    def _rows_early_hooks(self, targets, K):
        """Hooks for ``Transformer.backward_shared`` (shared forward, ``rows_only``): the decoder half of the rules
        (``ops.detr_decoder_rows``: 10 workgroups, latency-bound) and every encoder layer's head average (HBM-bound) need only
        slabs that are complete long before the backward ends, so they run on a side stream beside the rest of it.
        Returns ``(hooks, collect)``; ``collect()`` joins the side stream and returns ``(A, s, dec_word)``."""
        tr = self.model.transformer
        main = torch.cuda.current_stream()
        side = ops.side_stream(targets.device)          # (the same side stream as the value projections of the forward)
        got = {"A": [None] * len(tr.encoder.layers)}

        def pair(mod):
            return mod.get_attn().detach(), mod.get_attn_gradients().detach()

        def decoder_done():
            side.wait_stream(main)
            with torch.cuda.stream(side):
                got["s"], got["word"] = ops.detr_decoder_rows([pair(b.self_attn) for b in tr.decoder.layers],
                                                              [pair(b.multihead_attn) for b in tr.decoder.layers], targets,
                                                              shared_attn=True)

        def encoder_layer_done(i):
            side.wait_stream(main)
            with torch.cuda.stream(side):
                got["A"][i] = ops.avg_heads(*pair(tr.encoder.layers[i].self_attn), batch_size=K, shared_attn=True)

        def collect():
            main.wait_stream(side)
            for t in got["A"] + [got["s"], got["word"]]:
                t.record_stream(main)                  # made on the side stream, read (and released) on this one
            return got["A"], got["s"], got["word"]

        return {"decoder_done": decoder_done, "encoder_layer_done": encoder_layer_done}, collect

    def _rows_only_rules(self, encoder_blocks, decoder_blocks, targets, K, shared, pair, check_diag, early=None):
        """Rules 6 / 7 / 10 of ``generate_ours`` (DETR/modules/ExplanationGenerator.py:110-140) for ONE row per sample.

        With ``A_l`` the head-averaged encoder maps, ``B_l`` the decoder self-attention ones, ``C_l`` the cross-attention
        ones and ``N(.)`` = ``handle_residual``:   ``R_ii = (I+A_6)...(I+A_1)``,  ``R_qq^(l) = (I+B_l) R_qq^(l-1)``,
        ``R_qi^(l) = (I+B_l) R_qi^(l-1) + scrub( N(R_qq^(l))^T C_l N(R_ii) )``.  Row t of the final ``R_qi`` is therefore

            sum_l  u_l N(R_qq^(l))^T C_l  N(R_ii),      u_l = e_t^T (I+B_L)...(I+B_(l+1))   (a row vector, top-down)
          =  s N(R_ii)  with  s = sum_l u_l N(R_qq^(l))^T C_l,
          =  (s / rho) R_ii - s / rho + s,               rho = R_ii 1 - 1   (the row sums eq. 8-9 divides by),

        and both ``R_ii 1`` (bottom-up) and ``v R_ii`` (top-down) are chains of mat-vecs with the ``A_l``: the 950^3 products
        and the ``[Q, Ni]`` state never exist.  NaN policy of the reference (``R_sq_addition[isnan] = 0``): an addition
        holding one NaN holds only NaNs (every element sums over the offending row), so a layer whose factors contain a
        NaN contributes nothing, and a sample whose ``N(R_ii)`` is not finite (an all-zero row of ``R_ii - I``) returns
        zeros.  ``diag(R_ii - I) >= 0`` cannot be read without ``R_ii``; its entries are sums of products of clamped
        (>= 0) maps, so the assert can only fire on NaN, which ``min(rho) >= 0`` detects as well: that word stands in."""
        if not (self.apply_self_in_rule_10 and self.normalize_self_attention):
            raise NotImplementedError("rows_only covers the default rule set (normalize_self_attention, apply_self_in_rule_10)")
        dev = targets.device
        if early is not None:                                          # produced beside the backward (``_rows_early_hooks``)
            A, s_early, dec_word_early = early
        else:
            A = [ops.avg_heads(*pair(blk.self_attn), batch_size=K, shared_attn=shared) for blk in encoder_blocks]  # [K, Ni, Ni]
        n_img = A[0].shape[-1]
        # rho = R_ii 1 - 1 carried as the deviation e from 1 (e <- e + A (1 + e)): forming R_ii 1 and subtracting 1 at the
        # end would cancel 2-3 digits of rho (the matrix route has that noise in diag(R_ii) - 1 as well)
        e = torch.zeros(K, n_img, device=dev)
        for a in A:                                                    # bottom-up
            e = ops.chain_matvec(a, e + 1.0, base=e)
        rho = e.reshape(K, 1, n_img)
        # decoder half: B_l, R_qq^(l), N(R_qq^(l)), w_l = u_l N^T, z_l = w_l C_l and the NaN policy for ALL layers in three
        # launches (K2-DETR, csrc/detr_rows_kernels.hip) -- no C_l, no R_qq stack, no per-layer torch ops
        if early is not None:
            s, dec_word = s_early, dec_word_early
        else:
            s, dec_word = ops.detr_decoder_rows([pair(blk.self_attn) for blk in decoder_blocks],
                                                [pair(blk.multihead_attn) for blk in decoder_blocks], targets, shared_attn=shared)
        s = s.reshape(K, 1, n_img)
        words = [dec_word]
        v = s / rho
        v2 = v.reshape(K, n_img)
        d = torch.zeros_like(v2)                                       # d = v R_ii - v, top-down: d <- d + (v + d) A
        for a in reversed(A):
            d = ops.chain_vecmat(v2 + d, a, base=d)
        out = d.reshape(K, 1, n_img) + s
        finite = torch.isfinite(v).flatten(1).all(1).reshape(K, 1, 1)
        out = torch.where(finite & ~torch.isnan(out), out, torch.zeros_like(out))
        self.R_i_i = self.R_q_q = self.R_q_i = None
        self.diag_min = torch.cat(words + [rho.min().reshape(1)]).min()
        if check_diag is True:
            assert self.diag_min.item() >= 0
        return out.reshape(1, 1, K, n_img).detach()

    def generate_ours_multi(self, img, target_indices, index=None, normalize_self_attention=True,
                            apply_self_in_rule_10=True, share_forward=True, check_diag=True, rows_only=False):
        """All kept queries of one image in ONE pass.

        Equal to ``torch.cat([generate_ours(img, t, index, use_lrp=False, ...) for t in target_indices], dim=2)`` --
        the loop ``DETR/mask_generator.py:90-110`` runs, one full forward + backward + rule schedule per kept query --
        but the image is replicated into a batch of K, ONE backward carries the K one-hot seeds (sample k's logits
        depend on sample k only, so its slabs hold exactly the single-query gradients), and every rule runs batched:
        the encoder chain is one launch for all K, each decoder block is one rules-6+7 launch, one head-average launch
        and two batched MFMA matmuls; ``handle_residual(R_i_i)`` is hoisted out of the decoder loop (R_i_i does not
        change there).  ``img``: ``[1, C, h, w]`` float features.  Returns ``[1, 1, K, Ni]``.

        ``share_forward`` (default on, used when the body offers ``forward_shared`` / ``backward_shared``, i.e.
        ``detr_model.DETRFromFeatures``): the K replicas of the image have identical activations and differ only in
        their upstream gradients, so the forward runs ONCE at batch 1 and the backward -- the same chain of
        vector-Jacobian products, written out by hand -- runs at batch K; every block then holds ONE probability slab
        and K gradient slabs, which the rule kernels read with batch stride 0.  Same results to fp32 rounding.

        ``check_diag``: the reference asserts ``diag(R - I) >= 0`` inside every ``handle_residual`` (a device->host
        read each, 7 per call here).  ``True``: the 7 device words are reduced and asserted ONCE at the end; ``"defer"``:
        the reduced word is left in ``self.diag_min`` and nothing synchronises (``GraphedGenerateOursMulti``).

        ``rows_only=True``: the call returns ROW ``targets[k]`` of sample k's ``R_q_i`` and nothing else reads the
        matrices, so the rules are applied to row VECTORS (``_rows_only_rules``): no ``R_i_i`` (6 products of 950^3 per
        sample), no ``[Q, Ni]`` state, no rule-10 matmuls -- mat-vecs with the head-averaged maps instead.  Same result
        up to fp32 summation order; ``self.R_i_i`` / ``R_q_q`` / ``R_q_i`` are then ``None``.
        """
        self.use_lrp = False
        self.normalize_self_attention = normalize_self_attention
        self.apply_self_in_rule_10 = apply_self_in_rule_10
        targets = torch.as_tensor(target_indices, device=img.device).reshape(-1)
        K = targets.numel()
        if img.shape[0] != 1:
            raise ValueError("generate_ours_multi explains the queries of ONE image (got batch %d)" % img.shape[0])
        rows = torch.arange(K, device=img.device)
        shared = bool(share_forward and K > 1 and hasattr(self.model, "forward_shared"))
        early = None
        if shared:
            with torch.no_grad():
                logits, state = self.model.forward_shared(img, K)                                     # [1, Q, C+1]
                own = logits[0].index_select(0, targets)[:, :-1].argmax(dim=-1)
                if index is None:
                    index = own
                else:                                  # entries < 0 of a given index tensor mean "arg-max" (graph slots)
                    index = torch.as_tensor(index, device=img.device).reshape(-1)
                    index = torch.where(index < 0, own, index)
                # one-hot seeds by scatter (an index_put with tensor indices cannot be captured into a hipGraph)
                n_cls = logits.shape[-1]
                one_hot = torch.zeros(K, logits.shape[1] * n_cls, dtype=logits.dtype, device=img.device)
                one_hot.scatter_(1, (targets * n_cls + index).reshape(K, 1), 1.0)
                one_hot = one_hot.view(K, logits.shape[1], n_cls)
                early = self._rows_early_hooks(targets, K) if rows_only and self.overlap_rules else None
                if early is None:
                    self.model.backward_shared(state, one_hot)
                else:
                    self.model.backward_shared(state, one_hot, early[0])
        else:
            batch = img.expand(K, *img.shape[1:])
            outputs = rules.forward_for_backward(self.model, lambda: self.model(batch)["pred_logits"])   # [K, Q, C+1]
            if index is None:
                index = outputs[rows, targets, :-1].argmax(dim=-1)
            one_hot = torch.zeros_like(outputs)
            one_hot[rows, targets, index] = 1
            self.model.zero_grad()
            torch.sum(one_hot * outputs).backward(retain_graph=True)

        decoder_blocks = self.model.transformer.decoder.layers
        encoder_blocks = self.model.transformer.encoder.layers

        def pair(mod):
            return mod.get_attn().detach(), mod.get_attn_gradients().detach()

        if rows_only:
            return self._rows_only_rules(encoder_blocks, decoder_blocks, targets, K, shared, pair, check_diag,
                                         early[1]() if early is not None else None)
        enc = [pair(blk.self_attn) for blk in encoder_blocks]
        self.R_i_i = ops.relevancy_self_chain([a for a, _ in enc], [g for _, g in enc], K, shared_attn=shared)   # [K, Ni, Ni]
        n_img = self.R_i_i.shape[-1]
        n_q = decoder_blocks[0].self_attn.get_attn().shape[-1]
        self.R_q_q = torch.eye(n_q, device=img.device).repeat(K, 1, 1)
        self.R_q_i = torch.zeros(K, n_q, n_img, device=img.device)
        use_self = apply_self_in_rule_10
        R_ii_hat = self.R_i_i
        diag_words = []
        if use_self and normalize_self_attention:
            R_ii_hat, word = ops.handle_residual(self.R_i_i, check_diag="defer")
            diag_words.append(word)
        for blk in decoder_blocks:
            a, g = pair(blk.self_attn)
            self.R_q_q, self.R_q_i = ops.relevancy_self_chain([a], [g], K, R_init=self.R_q_q, R_sq_init=self.R_q_i,
                                                              shared_attn=shared)
            cam = ops.avg_heads(*pair(blk.multihead_attn), batch_size=K, shared_attn=shared)         # [K, Q, Ni]
            if not use_self:                                   # ablation: the addition is the cross-attention map
                self.R_q_i = self.R_q_i + cam
                continue
            R_qq_hat = self.R_q_q
            if normalize_self_attention:
                R_qq_hat, word = ops.handle_residual(self.R_q_q, check_diag="defer")
                diag_words.append(word)
            # R_q_i += nan_to_zero( R_qq_hat^T . (cam . R_ii_hat) )      (rule 10, DETR/...:33-43)
            addition = ops.matmul(R_qq_hat, ops.matmul(cam, R_ii_hat), trans_a=True, nan_to_zero=True)
            self.R_q_i = self.R_q_i + addition
        self.diag_min = torch.cat(diag_words).min() if diag_words else None
        if check_diag is True and self.diag_min is not None:
            assert self.diag_min.item() >= 0          # the reference's handle_residual asserts, one read for all of them
        # row targets[k] of sample k (capture-safe: one index_select on the flattened [K * Q, Ni] matrix)
        picked = self.R_q_i.reshape(K * n_q, n_img).index_select(0, rows * n_q + targets)
        return picked.reshape(1, 1, K, n_img).detach()

    # ------------------------------------------------------------------ baselines on the same slabs
    def generate_raw_attn(self, img, target_index):
        """Reference :225-238: head-mean of the last decoder cross-attention."""
        self.model(img)
        cam_q_i = self.model.transformer.decoder.layers[-1].multihead_attn.get_attn().detach()
        cam_q_i = cam_q_i.reshape(-1, cam_q_i.shape[-2], cam_q_i.shape[-1]).mean(dim=0)
        self.R_q_i = cam_q_i
        return self.R_q_i.unsqueeze_(0)[:, target_index, :].unsqueeze_(0)

    def generate_rollout(self, img, target_index):
        """Reference :240-273: rollout of the encoder / decoder self-attention, joined through the last cross-attention."""
        self.model(img)
        decoder_blocks = self.model.transformer.decoder.layers
        encoder_blocks = self.model.transformer.encoder.layers
        cams_image = [blk.self_attn.get_attn().detach().mean(dim=0) for blk in encoder_blocks]
        cams_queries = [blk.self_attn.get_attn().detach().mean(dim=0) for blk in decoder_blocks]
        # the reference indexes ``shape[1]`` of ``[N, N]`` maps here, which only works because they are square
        self.R_i_i = compute_rollout_attention(cams_image)
        self.R_q_q = compute_rollout_attention(cams_queries)
        cam_q_i = decoder_blocks[-1].multihead_attn.get_attn().detach()
        cam_q_i = cam_q_i.reshape(-1, cam_q_i.shape[-2], cam_q_i.shape[-1]).mean(dim=0)
        self.R_q_i = ops.matmul(self.R_q_q, ops.matmul(cam_q_i, self.R_i_i), trans_a=True)
        return self.R_q_i.unsqueeze_(0)[:, target_index, :].unsqueeze_(0)

    def gradcam(self, cam, grad):
        return rules.gradcam(cam, grad)

    def generate_attn_gradcam(self, img, target_index, index=None):
        """Reference :282-305."""
        outputs = _logits_for_backward(self.model, img)
        _one_hot_backward(self.model, outputs, target_index, index)
        last = self.model.transformer.decoder.layers[-1].multihead_attn
        self.R_q_i = self.gradcam(last.get_attn().detach(), last.get_attn_gradients().detach())
        return self.R_q_i.unsqueeze_(0)[:, target_index, :].unsqueeze_(0)

    def generate_transformer_att(self, img, target_index, index=None):
        """Reference :64-108: rule 5 on the LRP cam of the LAST decoder cross-attention (needs the body's ``relprop``)."""
        _require_relprop(self.model)
        outputs = _logits_for_backward(self.model, img)
        index, one_hot_vector = _one_hot_backward(self.model, outputs, target_index, index)
        _relprop(self.model, one_hot_vector, target_index, index)
        last = self.model.transformer.decoder.layers[-1].multihead_attn
        self.R_q_i = avg_heads(last.get_attn_cam().detach(), last.get_attn_gradients().detach())
        return self.R_q_i.unsqueeze_(0)[:, target_index, :].unsqueeze_(0)

    def generate_partial_lrp(self, img, target_index, index=None):
        """Reference :197-223: head-mean of the last cross-attention's LRP cam, min-max normalised (no backward)."""
        _require_relprop(self.model)
        outputs = self.model(img)["pred_logits"]
        index, one_hot_vector = _one_hot_backward(self.model, outputs, target_index, index, backward=False)
        _relprop(self.model, one_hot_vector, target_index, index)
        cam_q_i = self.model.transformer.decoder.layers[-1].multihead_attn.get_attn_cam().detach()
        cam_q_i = cam_q_i.reshape(-1, cam_q_i.shape[-2], cam_q_i.shape[-1]).mean(dim=0)
        self.R_q_i = (cam_q_i - cam_q_i.min()) / (cam_q_i.max() - cam_q_i.min())
        return self.R_q_i.unsqueeze_(0)[:, target_index, :].unsqueeze_(0)


class GeneratorAlbationNoAgg:
    """No-aggregation ablation (reference :310-403): every ``+=`` of ``generate_ours`` becomes ``=``."""

    def __init__(self, model):
        self.model = model
        self.model.eval()

    def forward(self, input_ids, attention_mask):
        return self.model(input_ids, attention_mask)

    def handle_self_attention_image(self, blocks):
        for blk in blocks:
            cam = avg_heads(_cam(blk.self_attn, self.use_lrp), blk.self_attn.get_attn_gradients().detach())
            self.R_i_i = ops.matmul(cam, self.R_i_i)

    def handle_co_attn_self_query(self, block):
        cam = avg_heads(_cam(block.self_attn, self.use_lrp), block.self_attn.get_attn_gradients().detach())
        self.R_q_q, self.R_q_i = apply_self_attention_rules(self.R_q_q, self.R_q_i, cam)

    def handle_co_attn_query(self, block):
        cam_q_i = avg_heads(_cam(block.multihead_attn, self.use_lrp), block.multihead_attn.get_attn_gradients().detach())
        self.R_q_i = apply_mm_attention_rules(self.R_q_q, self.R_i_i, cam_q_i,
                                              apply_normalization=self.normalize_self_attention,
                                              apply_self_in_rule_10=self.apply_self_in_rule_10)

    def generate_ours_abl(self, img, target_index, index=None, use_lrp=False, normalize_self_attention=False,
                          apply_self_in_rule_10=True):
        self.use_lrp = use_lrp
        self.normalize_self_attention = normalize_self_attention
        self.apply_self_in_rule_10 = apply_self_in_rule_10          # reference :342, forwarded to rule 10 (:347-349)
        outputs = _logits_for_backward(self.model, img)
        index, one_hot_vector = _one_hot_backward(self.model, outputs, target_index, index)
        if use_lrp:
            _relprop(self.model, one_hot_vector, target_index, index)
        decoder_blocks = self.model.transformer.decoder.layers
        encoder_blocks = self.model.transformer.encoder.layers
        ref = encoder_blocks[0].self_attn.get_attn()
        image_bboxes = ref.shape[-1]
        queries_num = decoder_blocks[0].self_attn.get_attn().shape[-1]
        self.R_i_i = torch.eye(image_bboxes, image_bboxes, device=ref.device)
        self.R_q_q = torch.eye(queries_num, queries_num, device=ref.device)
        self.R_q_i = torch.zeros(queries_num, image_bboxes, device=ref.device)
        self.handle_self_attention_image(encoder_blocks)
        for blk in decoder_blocks:
            self.handle_co_attn_self_query(blk)
            self.handle_co_attn_query(blk)
        aggregated = self.R_q_i.unsqueeze_(0)
        return aggregated[:, target_index, :].unsqueeze_(0).detach()


class MaskGenerator:
    """The per-image core of ``DETR/mask_generator.py:40-125`` (``MaskGenerator.get_panoptic``) without the COCO /
    visualisation plumbing: keep the confident queries, explain them, binarise each relevancy map with Otsu.

    ``get_masks(img, method)`` -> ``(masks [1, Q, h, w], keep [Q] bool)``; rows of queries that are not kept hold -1
    like the reference's buffer.  For the three rule-based methods every kept query goes through
    ``Generator.generate_ours_multi`` in one pass and all Otsu thresholds are one launch
    (``postprocess.otsu_masks``); the remaining methods run the reference's per-query loop.
    """

    _BATCHED = {"ours_no_lrp": {}, "ablation_no_self_in_10": {"apply_self_in_rule_10": False},
                "ours_no_lrp_no_norm": {"normalize_self_attention": False}}

    _PER_QUERY = ("ablation_no_aggregation", "ours_with_lrp", "raw_attn", "rollout", "attn_gradcam", "transformer_att", "partial_lrp")

    def __init__(self, model, threshold=0.5, graph_slots=None, max_graphs=4):
        """``graph_slots`` (e.g. 16): run the batched methods through ``GraphedGenerateOursMulti`` with that many target
        slots (captured on the first image of a given feature-map size; the evaluator's images are resized to a common
        size per batch, ``DETR/datasets/coco.py:138-140``).  Each captured pass owns a private memory pool and its pinned
        slabs, so at most ``max_graphs`` (method, feature-map size) pairs are kept, least recently used evicted."""
        self.gen = Generator(model)
        self.abl = GeneratorAlbationNoAgg(model)
        self.model = model
        self.threshold = threshold
        if graph_slots is not None and graph_slots < 2:
            raise ValueError("graph_slots must be >= 2 (the K-slot pass shares ONE forward between its slots)")
        self.graph_slots = graph_slots
        self.max_graphs = max_graphs
        self._graphs = collections.OrderedDict()     # (method, feature shape) -> captured pass, least recently used first
        self._diag_running = None                    # smallest handle_residual word of the batched passes not checked yet
        self.last_kept = 0

    def _fold_diag(self, word):
        """Keep the smallest ``handle_residual`` word of the batched passes on the device (NaN propagates: ``minimum``)."""
        if word is not None:
            w = word.detach().reshape(1).to(torch.float32)
            self._diag_running = w.clone() if self._diag_running is None else torch.minimum(self._diag_running, w)

    def check_diag(self):
        """The reference's ``assert diag.min() >= 0`` for the passes not checked yet (call after the last image)."""
        if self._diag_running is not None:
            word, self._diag_running = float(self._diag_running), None
            assert word >= 0, "handle_residual: diag(R - I) < 0 (or NaN) in a relevancy pass"

    def _per_query(self, img, idx, method):
        """The reference's per-query dispatch (DETR/mask_generator.py:91-113) for the methods that are not batched here; the
        LRP ones run the body's ``relprop`` (``detr_model`` has it; a body without one raises ``NotImplementedError``)."""
        if method == "ablation_no_aggregation":
            return self.abl.generate_ours_abl(img, idx, use_lrp=False, normalize_self_attention=False)
        if method == "ours_with_lrp":
            return self.gen.generate_ours(img, idx, use_lrp=True)
        fn = {"raw_attn": self.gen.generate_raw_attn, "rollout": self.gen.generate_rollout,
              "attn_gradcam": self.gen.generate_attn_gradcam, "transformer_att": self.gen.generate_transformer_att,
              "partial_lrp": self.gen.generate_partial_lrp}.get(method)
        return None if fn is None else fn(img, idx)     # unknown method: the caller prints and returns None like the reference

    def get_masks(self, img, method="ours_no_lrp", outputs=None):
        """``outputs``: the body's output dict for ``img`` when the caller has already run the forward (an evaluator
        that looks at the confidences first); default: run it here, like ``mask_generator.py:47``."""
        from . import postprocess
        if outputs is None:
            with torch.no_grad():
                outputs = self.model(img)
        h, w = self.model.spatial_dim
        probas = outputs["pred_logits"].softmax(-1)[0, :, :-1]
        keep = probas.max(-1).values > self.threshold               # (``threshold`` may be a 0-dim device tensor)
        # ONE device -> host read per image: the keep mask (its size decides how many passes follow) together with the
        # ``handle_residual`` word left by the passes of the PREVIOUS images (the reference asserts it inside every call; here
        # the batched passes leave it on the device and it is checked one image late, the last one by ``check_diag()``)
        word = self._diag_running if self._diag_running is not None else torch.zeros(1, device=img.device)
        host = torch.cat((keep.to(torch.float32), word.reshape(1).to(torch.float32))).cpu()
        assert float(host[-1]) >= 0, "handle_residual: diag(R - I) < 0 (or NaN) in a relevancy pass of an earlier image"
        kept = torch.nonzero(host[:-1]).reshape(-1).to(img.device, non_blocking=True)
        self.last_kept = int(kept.numel())
        masks = torch.full((1, probas.shape[0], h, w), -1.0, device=img.device)
        if kept.numel() == 0:
            return masks, keep
        # the class each kept query is explained for = arg-max of THIS forward (mask_generator.py passes index=None and
        # the generator re-derives it from a second, identical forward)
        classes = outputs["pred_logits"][0, kept, :-1].argmax(dim=-1)
        if method in self._BATCHED and self.graph_slots and hasattr(self.model, "forward_shared"):
            key = (method, tuple(img.shape))
            if key not in self._graphs:
                while len(self._graphs) >= self.max_graphs:
                    self._graphs.popitem(last=False)
                self._graphs[key] = GraphedGenerateOursMulti(self.model, img, self.graph_slots, **self._BATCHED[method])
            self._graphs.move_to_end(key)
            cams = self._graphs[key](img, kept, index=classes, check=self._fold_diag)[0, 0]
        elif method in self._BATCHED:
            cams = self.gen.generate_ours_multi(img, kept, index=classes, check_diag="defer",
                                                **self._BATCHED[method])[0, 0]                             # [K, Ni]
            self._fold_diag(self.gen.diag_min)
        else:
            if method not in self._PER_QUERY:
                # the reference's error convention (DETR/mask_generator.py:111-113): a message and ``None``, reached -- as there,
                # inside the loop over the kept queries -- only when the image has a kept query
                print("please provide a valid explainability method")
                return None
            cams = torch.cat([self._per_query(img, idx.reshape(1), method).reshape(1, -1) for idx in kept])
        masks[0, kept] = postprocess.otsu_masks(cams).reshape(-1, h, w)
        return masks, keep


class GraphedGenerateOursMulti:
    """``Generator.generate_ours_multi`` (shared forward, all K kept queries of an image in one pass) captured into a
    hipGraph for a FIXED number of target slots ``K`` and replayed: the pass is ~600 launches for ~3 ms of GPU work, i.e.
    host-bound when run eagerly.  An image with fewer kept queries fills the spare slots with repeats of its last target
    (their rows are dropped); one with more is processed in chunks of ``K``.  The ``handle_residual`` asserts become ONE
    device word read per call.

        run = GraphedGenerateOursMulti(model, example_features, K=16)
        maps = run(features, kept_query_indices)        # [1, 1, len(kept), Ni], == Generator(model).generate_ours_multi(...)
    """

    def __init__(self, model, example_img, K=16, normalize_self_attention=True, apply_self_in_rule_10=True, warmup=2,
                 rows_only=None):
        if not hasattr(model, "forward_shared"):
            raise ValueError("GraphedGenerateOursMulti needs a body with forward_shared / backward_shared (detr_model)")
        if K < 2:
            raise ValueError("GraphedGenerateOursMulti: K >= 2 target slots (one forward is shared between the slots; a "
                             "single slot would capture the autograd route, which does not replay)")
        self.K = K
        self.img = example_img.clone()
        self.targets = torch.zeros(K, dtype=torch.long, device=example_img.device)
        # class per slot: -1 = "arg-max of this pass's own logits" (the reference's index=None), else the given class.
        # A caller that already ran the forward (MaskGenerator) hands in ITS arg-max: two forwards of the same image can
        # round differently (library GEMM selection inside / outside a capture) and flip a near-tie between classes.
        self.index = torch.full((K,), -1, dtype=torch.long, device=example_img.device)
        self.gen = Generator(model)
        # only rows are returned: the row-vector form of the rules (no 950^3 products) whenever the default rules are on
        if rows_only is None:
            rows_only = bool(normalize_self_attention and apply_self_in_rule_10)
        kw = dict(normalize_self_attention=normalize_self_attention, apply_self_in_rule_10=apply_self_in_rule_10,
                  check_diag="defer", rows_only=rows_only)
        self._call = lambda: self.gen.generate_ours_multi(self.img, self.targets, index=self.index, **kw)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._call()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with ops.graph_capture(self.graph):
            self.out = self._call()
            self.diag_min = self.gen.diag_min
        # the graph holds raw addresses of the attention modules' slabs and of scratch buffers that were allocated by the
        # warm-up calls: keep them alive (an eager forward with another batch size replaces the modules' slabs)
        self._pinned = ops.pinned_state(model)

    def __call__(self, img, target_indices, index=None, check=True):
        targets = torch.as_tensor(target_indices, device=self.img.device).reshape(-1)
        n = targets.numel()
        classes = None if index is None else torch.as_tensor(index, device=self.img.device).reshape(-1)
        self.img.copy_(img)
        chunks = []
        for i in range(0, n, self.K):
            part = targets[i:i + self.K]
            self.targets[:part.numel()] = part
            self.index.fill_(-1)
            if classes is not None:
                self.index[:part.numel()] = classes[i:i + self.K]
            if part.numel() < self.K:
                self.targets[part.numel():] = part[-1]
                self.index[part.numel():] = self.index[part.numel() - 1]
            self.graph.replay()
            if callable(check):                       # the caller folds the device word into its own check (no sync here)
                check(self.diag_min)
            elif check and self.diag_min is not None:
                assert self.diag_min.item() >= 0
            chunks.append(self.out[:, :, :part.numel()].clone())
        return torch.cat(chunks, dim=2) if len(chunks) != 1 else chunks[0]
