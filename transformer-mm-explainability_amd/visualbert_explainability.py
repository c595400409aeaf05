"""VisualBERT (single-stream) relevancy -- ``VisualBERT/mmf/models/transformers/backends/ExplanationGenerator.py``
surface (``SelfAttentionGenerator``) on the HIP kernels.

``model`` is duck-typed as in the reference: ``model(input)['scores']``, ``model.model.bert.encoder.layer[i].attention.self``
with ``get_attn()`` / ``get_attn_gradients()`` -> ``[1, H, N, N]``; ``input['input_mask']`` gives the ``[CLS]``-row index
``input_mask.sum(1) - 2`` (reference :94-95).  Visualisation flags are accepted and ignored (cv2 drawing is not part of
the path).  LRP methods (``generate_transformer_att``, ``generate_partial_lrp``) run on ``get_attn_cam()``, filled by the
body's LRP pass ``model.relprop(one_hot, alpha=1)`` (BERT_ours.py:345-395, visual_bert.py:398-403);
``visualbert_model.VisualBERT.relprop`` is that pass (``bert_lrp.py``: closed-form rules around the HIP attention-core
kernels, 'vqa' pooling like the reference); a body without ``relprop`` raises ``NotImplementedError``.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops, rules

compute_rollout_attention = rules.compute_rollout_attention_batched


def _backward_on_answer(model, input, index, use_lrp=False, backward=True):
    if use_lrp and not hasattr(model, "relprop"):
        raise NotImplementedError(
            "transformer_att / partial_lrp read LRP attention cams (get_attn_cam) that the body's relprop() must produce; "
            "%s has no relprop().  Plug a body built on an LRP layer library (reference: BERT_ours.py)." % type(model).__name__)
    output = rules.forward_for_backward(model, lambda: model(input)["scores"])
    if index is None:
        index = np.argmax(output.cpu().data.numpy(), axis=-1)
    one_hot = torch.zeros_like(output)
    one_hot[0, index] = 1
    if backward:
        loss = torch.sum(one_hot * output)
        model.zero_grad()
        loss.backward(retain_graph=True)
    if use_lrp:
        model.relprop(one_hot.detach().clone(), alpha=1)


def save_visual_results(*_args, **_kwargs):
    """``save_visualization=True`` / ``save_visualization_per_token=True``: the reference calls ``save_visual_results(input, scores,
    method_name=...)`` here (VisualBERT/mmf/models/transformers/backends/ExplanationGenerator.py:58-66, :99-107, :129-131, :164-166,
    :182-184, :213-215) -- a function that file neither defines nor imports, so the reference itself raises ``NameError`` at this
    point, after the scores are computed.  This package writes no images either (cv2 is not a dependency); it fails the same way,
    loudly, instead of accepting the flag and doing nothing."""
    raise NameError("name 'save_visual_results' is not defined -- the reference's ExplanationGenerator.py calls it without defining "
                    "it; call the generator with save_visualization=False and render the returned scores yourself "
                    "(transformer_mm_explainability_amd.postprocess)")


class SelfAttentionGenerator:
    def __init__(self, model):
        self.model = model
        self.model.eval()
        self.use_tape = True       # generate_ours_batch: hand-written forward / backward of the body when it offers one

    def _blocks(self):
        return self.model.model.bert.encoder.layer

    def generate_ours(self, input, index=None, save_visualization=False, save_visualization_per_token=False):
        """Reference :68-107 -> ``[1, N]`` row of the answer token with its own column zeroed."""
        _backward_on_answer(self.model, input, index)
        blocks = self._blocks()
        attn = [blk.attention.self.get_attn()[0] for blk in blocks]            # cam[0] -> [H, N, N]
        grad = [blk.attention.self.get_attn_gradients()[0] for blk in blocks]
        cls_index = input["input_mask"].sum(1) - 2
        cls_per_token_score = ops.relevancy_chain_row(attn, grad, 1, cls_index)          # R[cls_index], [1, N]
        cls_per_token_score[:, cls_index] = 0
        if save_visualization or save_visualization_per_token:
            save_visual_results(input, cls_per_token_score, method_name='generate_ours')
        return cls_per_token_score

    def generate_ours_batch(self, input, index=None, _n_text=None):
        """B items with the same number of real text tokens in ONE forward + ONE backward + ONE chain launch.

        ``input``: the ``sample_list`` dict with batch-first tensors of B items whose ``input_mask`` rows have equal sums
        (the wrapper trims the text padding by that length, ``visual_bert.py:578-588``).  Samples are independent, so B
        one-hot seeds in one backward leave the per-sample gradients in the slabs and the chain kernel runs one
        (sample, layer group) per workgroup.  Returns ``[B, N]``: row ``b`` equals ``generate_ours`` of item ``b``.
        """
        mask = input["input_mask"]
        if _n_text is None:
            lengths = mask.sum(1)
            if int((lengths != lengths[0]).sum()) != 0:
                raise ValueError("generate_ours_batch needs items of equal text length (bucket them: sharding.length_buckets)")
            n_text = int(lengths[0])
        else:
            n_text = _n_text                       # (a captured replay: the length was fixed, and checked, at capture time)
        B = mask.shape[0]
        model = self.model.model                                     # VisualBERTForClassification
        ids, seg = input["input_ids"][:, :n_text], input["segment_ids"][:, :n_text]
        feats = input["image_feature_0"]
        text_mask = torch.ones(B, n_text, dtype=torch.long, device=ids.device)
        visual_mask = torch.ones(B, feats.shape[1], dtype=torch.long, device=ids.device)
        args = (ids, text_mask, torch.cat((text_mask, visual_mask), dim=-1), seg, feats, torch.zeros_like(visual_mask))
        if self.use_tape and hasattr(model, "forward_tape"):
            # hand-written forward / backward of the BERT stack (visualbert_model.forward_tape, bert_tape.py)
            output, state = model.forward_tape(*args)
            idx = output.argmax(dim=-1) if index is None else torch.as_tensor(index, device=output.device).reshape(-1)
            model.backward_tape(state, torch.zeros_like(output).scatter_(1, idx.reshape(-1, 1), 1.0))
        else:
            output = rules.forward_for_backward(self.model, lambda: model(*args)["scores"])
            idx = output.argmax(dim=-1) if index is None else torch.as_tensor(index, device=output.device).reshape(-1)
            one_hot = torch.zeros_like(output).scatter_(1, idx.reshape(-1, 1), 1.0)
            self.model.zero_grad()
            torch.sum(one_hot * output).backward(retain_graph=True)
        blocks = self._blocks()
        cls_index = n_text - 2
        scores = ops.relevancy_chain_row([blk.attention.self.get_attn() for blk in blocks],
                                         [blk.attention.self.get_attn_gradients() for blk in blocks], B, cls_index)    # [B, N]
        scores[:, cls_index] = 0
        return scores

    def generate_rollout(self, input, start_layer=0, save_visualization=False):
        """Reference :158-174: head-mean maps (``sum/H``), batched rollout WITHOUT row normalisation."""
        self.model(input)
        cams = []
        for blk in self._blocks():
            attn_heads = blk.attention.self.get_attn()
            cams.append((attn_heads.sum(dim=1) / attn_heads.shape[1]).detach())        # [1, N, N]
        rollout = compute_rollout_attention(cams, start_layer=start_layer)
        cls_index = input["input_mask"].sum(1) - 2
        cls_per_token_score = rollout[0, cls_index]
        cls_per_token_score[:, cls_index] = 0
        if save_visualization:
            save_visual_results(input, cls_per_token_score, method_name='generate_rollout')
        return cls_per_token_score

    def generate_raw_attn(self, input, save_visualization=False):
        """Reference :145-156: head-mean of the last layer."""
        self.model(input)
        cam = self._blocks()[-1].attention.self.get_attn()[0].mean(dim=0).unsqueeze(0)
        cls_index = input["input_mask"].sum(1) - 2
        cls_per_token_score = cam[0, cls_index]
        cls_per_token_score[:, cls_index] = 0
        if save_visualization:
            save_visual_results(input, cls_per_token_score, method_name='generate_raw_attn')
        return cls_per_token_score

    def generate_attn_gradcam(self, input, index=None, save_visualization=False):
        """Reference :176-215: GradCAM of the last layer, min-max normalised."""
        _backward_on_answer(self.model, input, index)
        sa = self._blocks()[-1].attention.self
        cam = rules.gradcam(sa.get_attn()[0], sa.get_attn_gradients()[0]).unsqueeze(0)
        cam = (cam - cam.min()) / (cam.max() - cam.min())
        cls_index = input["input_mask"].sum(1) - 2
        cls_per_token_score = cam[0, cls_index]
        cls_per_token_score[:, cls_index] = 0
        if save_visualization:
            save_visual_results(input, cls_per_token_score, method_name='generate_attn_gradcam')
        return cls_per_token_score

    def generate_transformer_att(self, input, index=None, start_layer=0, save_visualization=False,
                                 save_visualization_per_token=False):
        """Reference :24-66: rule 5 on the LRP cams, then the un-normalised batched rollout from ``start_layer``."""
        _backward_on_answer(self.model, input, index, use_lrp=True)
        cams = []
        for blk in self._blocks():
            sa = blk.attention.self
            cams.append(ops.avg_heads(sa.get_attn_cam()[0], sa.get_attn_gradients()[0], batch_size=1))     # [1, N, N]
        rollout = compute_rollout_attention(cams, start_layer=start_layer)
        cls_index = input["input_mask"].sum(1) - 2
        cls_per_token_score = rollout[0, cls_index]
        cls_per_token_score[:, cls_index] = 0
        if save_visualization or save_visualization_per_token:
            save_visual_results(input, cls_per_token_score, method_name='generate_transformer_att')
        return cls_per_token_score

    def generate_partial_lrp(self, input, index=None, save_visualization=False):
        """Reference :109-130: head-mean of the last layer's LRP cam, min-max normalised (no backward)."""
        _backward_on_answer(self.model, input, index, use_lrp=True, backward=False)
        cam = self._blocks()[-1].attention.self.get_attn_cam()[0].mean(dim=0).unsqueeze(0)
        cam = (cam - cam.min()) / (cam.max() - cam.min())
        cls_index = input["input_mask"].sum(1) - 2
        cls_per_token_score = cam[0, cls_index]
        cls_per_token_score[:, cls_index] = 0
        if save_visualization:
            save_visual_results(input, cls_per_token_score, method_name='generate_partial_lrp')
        return cls_per_token_score


class GraphedGenerateOursBatch:
    """``SelfAttentionGenerator.generate_ours_batch`` captured once into a hipGraph and replayed (the batched explain pass is
    a few hundred launches of a few microseconds each, i.e. bound by the host when run eagerly).  Batch size, number of
    regions and the number of real text tokens are fixed at construction; new ``input_ids`` / ``segment_ids`` /
    ``image_feature_0`` values are copied into the captured buffers.

        run = GraphedGenerateOursBatch(model, example_sample_list)      # B items of equal text length
        scores = run(sample_list)                                        # [B, N], == generate_ours_batch(sample_list)
    """

    KEYS = ("input_ids", "input_mask", "segment_ids", "image_feature_0")

    def __init__(self, model, example, index=None, warmup=2):
        lengths = example["input_mask"].sum(1)
        if int((lengths != lengths[0]).sum()) != 0:
            raise ValueError("GraphedGenerateOursBatch needs items of equal text length")
        self.n_text = int(lengths[0])
        self.static = {k: example[k].clone() for k in self.KEYS}
        self.static_index = None if index is None else torch.as_tensor(index, device=example["input_ids"].device).clone()
        self.gen = SelfAttentionGenerator(model)
        self._call = lambda: self.gen.generate_ours_batch(self.static, self.static_index, _n_text=self.n_text)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._call()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with ops.graph_capture(self.graph):
            self.output = self._call()
        self._pinned = ops.pinned_state(model)

    def __call__(self, input=None, index=None):
        if input is not None:
            if int(input["input_mask"].sum(1)[0]) != self.n_text:
                raise ValueError("this graph was captured for %d text tokens" % self.n_text)
            for k in self.KEYS:
                if input[k].shape != self.static[k].shape:
                    raise ValueError("%s: %s, captured for %s" % (k, tuple(input[k].shape), tuple(self.static[k].shape)))
                self.static[k].copy_(input[k])
        if index is not None:
            if self.static_index is None:
                raise ValueError("the graph was captured with index=None (arg-max answers)")
            self.static_index.copy_(torch.as_tensor(index))
        self.graph.replay()
        return self.output
