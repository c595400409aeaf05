"""Reference-style hooked attention modules backed by the HIP capture op.

The reference patches every model family's attention class with the same six accessors
(``save_attn / get_attn / save_attn_gradients / get_attn_gradients / save_attn_cam / get_attn_cam``) fed by a Python
forward hook and a tensor backward hook.  These classes keep that surface but the tensors behind ``get_attn()`` /
``get_attn_gradients()`` are the slabs the HIP kernels write (no hook, no copy):

  * ``MultiheadAttention``  -- DETR/modules/layers.py:666-768: separate q/k/v projections, ``q*d^-0.5`` first,
    inputs ``[T, B, E]``, masks accepted and ignored (exactly like the reference, layers.py:728-756),
    ``get_attn()`` -> ``[B*H, T, S]``.  Loads the reference's state dict (incl. packed ``in_proj_*`` checkpoints).
  * ``BertStyleAttention``  -- lxmert/lxmert/src/lxmert_lrp.py:322-420 (``LxmertAttention``) and
    VisualBERT/.../BERT_ours.py:234-343 (``BertSelfAttention``): ``query/key/value`` Linear, ``scores/sqrt(d) + mask``,
    inputs ``[B, N, E]``, ``get_attn()`` -> ``[B, H, Nq, Nk]`` (captured before dropout; eval mode => identical).

``save_attn_cam`` / ``get_attn_cam`` are the plain slot of the reference (DETR/modules/layers.py:699-703): an LRP pass
(``model.relprop``) stores its per-head relevance there and the rule kernels read it like any other cam.
``MultiheadAttention.relprop`` (layers.py:770-801) produces it: closed-form Linear rules (``lrp.py``) around the HIP
attention-core kernels (``csrc/attention_lrp.hip``).  The BERT-style module keeps a tape for ``bert_lrp.attention_relprop``
(the ``relprop`` of the LXMERT / VisualBERT bodies); a caller-supplied LRP pass may still fill the slot; reading an empty slot
raises.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import _lib, ops
from .capture import attention_capture


class _SlabOwner(nn.Module):
    def __init__(self):
        super().__init__()
        self._probs = None
        self._grads = None
        self.attn = None
        self.attn_gradients = None
        self.attn_cam = None

    def _slabs(self, B, H, Nq, Nk, device):
        shape = (B, H, Nq, Nk)
        if self._probs is None or tuple(self._probs.shape) != shape or self._probs.device != device:
            self._probs = torch.empty(shape, dtype=torch.float32, device=device)
        if self._grads is None or tuple(self._grads.shape) != shape or self._grads.device != device:
            self._grads = torch.empty(shape, dtype=torch.float32, device=device)   # (shared-forward mode sizes it by K)
        return self._probs, self._grads

    # the reference's accessor surface (DETR/modules/layers.py:693-709, lxmert_lrp.py:356-372, BERT_ours.py:266-282)
    def save_attn(self, attn):
        self.attn = attn

    def get_attn(self):
        return self.attn

    def save_attn_gradients(self, attn_gradients):
        self.attn_gradients = attn_gradients

    def get_attn_gradients(self):
        return self.attn_gradients

    def save_attn_cam(self, cam):
        self.attn_cam = cam

    def get_attn_cam(self):
        if self.attn_cam is None:
            raise NotImplementedError("no LRP attention cam has been saved on this module: an LRP pass (model.relprop) "
                                      "must call save_attn_cam first; use the *_no_lrp methods otherwise")
        return self.attn_cam


class MultiheadAttention(_SlabOwner):
    """DETR-style hooked MHA on the HIP capture op (see module docstring)."""

    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        self.embed_dim, self.num_heads, self.head_dim = embed_dim, num_heads, embed_dim // num_heads
        self.q_proj = nn.Linear(embed_dim, embed_dim)
        self.k_proj = nn.Linear(embed_dim, embed_dim)
        self.v_proj = nn.Linear(embed_dim, embed_dim)
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=True)
        self.dropout_p = dropout
        self._register_load_state_dict_pre_hook(self._split_packed_in_proj)

    @staticmethod
    def _split_packed_in_proj(state_dict, prefix, *args):
        """Accept ``nn.MultiheadAttention`` checkpoints (packed ``in_proj_weight``), like layers.py:711-726."""
        if prefix + "in_proj_weight" in state_dict:
            w, b = state_dict.pop(prefix + "in_proj_weight"), state_dict.pop(prefix + "in_proj_bias")
            e = w.shape[1]
            for i, n in enumerate(("q_proj", "k_proj", "v_proj")):
                state_dict[prefix + n + ".weight"] = w[i * e:(i + 1) * e]
                state_dict[prefix + n + ".bias"] = b[i * e:(i + 1) * e]

    def forward(self, query, key, value, key_padding_mask=None, need_weights=True, attn_mask=None):
        if self.training and self.dropout_p > 0:
            raise _lib.MMXError("the capture op implements eval-mode attention (the reference generators call .eval())")
        T, B, E = query.shape
        S = key.shape[0]
        H, D = self.num_heads, self.head_dim
        # [T, B, E] -> [B, T, H, D] strided views; the kernel takes (batch, head, token) strides
        q = self.q_proj(query).view(T, B, H, D).permute(1, 0, 2, 3)
        k = self.k_proj(key).view(S, B, H, D).permute(1, 0, 2, 3)
        v = self.v_proj(value).view(S, B, H, D).permute(1, 0, 2, 3)
        probs, grads = self._slabs(B, H, T, S, query.device)
        o = attention_capture(q, k, v, probs, grads, float(D) ** -0.5, mask=None, scale_mode=_lib.SCALE_Q_FIRST)
        self.save_attn(probs.view(B * H, T, S))
        self.save_attn_gradients(grads.view(B * H, T, S))   # filled by the backward kernel
        # what an LRP pass needs of this forward (references only; the reference keeps the same tensors alive through its
        # forward hooks, DETR/modules/layers.py:17-30): inference under no_grad does not pin them
        self._lrp_tape = None
        if torch.is_grad_enabled():
            bf = lambda t: t.detach().permute(1, 0, 2)                          # noqa: E731  [N, B, E] -> [B, N, E]
            self._lrp_tape = dict(query=bf(query), key=bf(key), value=bf(value), q=q.detach(), k=k.detach(), v=v.detach(),
                                  o=o.detach(), probs=probs, scale=float(D) ** -0.5)
        o = o.permute(1, 0, 2, 3).reshape(T, B, E)
        return self.out_proj(o)

    def relprop(self, cam_attn_output, alpha=1, **kwargs):
        """``MultiheadAttention.relprop`` (DETR/modules/layers.py:770-801): ``cam_attn_output [T, B, E]`` -> ``(cam_q [T, B, E],
        cam_k [S, B, E], cam_v [S, B, E])``; the per-head relevance of the probabilities is stored with ``save_attn_cam``
        (``[B*H, T, S]``, layers.py:776).  The Linear rules are closed-form matrix products (``lrp.py``), the attention core
        (both ``einsum`` relprops) runs in the HIP kernels of ``csrc/attention_lrp.hip``."""
        from . import lrp
        t = getattr(self, "_lrp_tape", None)
        if t is None:
            raise _lib.MMXError("relprop needs the activations of a forward run with gradients enabled")
        core = lambda cam_o: ops.attn_relprop(t["q"], t["k"], t["v"], t["probs"], t["o"], cam_o, t["scale"],   # noqa: E731
                                              _lib.SCALE_Q_FIRST, layout="bnhd")
        with torch.no_grad():
            cam_q, cam_k, cam_v, cam_p = lrp.mha_relprop(
                cam_attn_output.permute(1, 0, 2), t,
                (self.q_proj.weight, self.k_proj.weight, self.v_proj.weight, self.out_proj.weight), core)
        B, H, T, S = cam_p.shape
        self.save_attn_cam(cam_p.view(B * H, T, S))
        return cam_q.permute(1, 0, 2), cam_k.permute(1, 0, 2), cam_v.permute(1, 0, 2)


    overlap_value_proj = True     # shared-forward mode: the value projection on a side stream beside the packed q / k GEMM

    # ---- shared-forward mode (one forward at batch 1, K upstream gradients): batch-first tensors
    def forward_shared(self, query, key, value, batch, kv=None):
        """``query [1, T, E]``, ``key`` / ``value [1, S, E]``.  Returns ``(out [1, T, E], tape)``; ``get_attn()`` then is
        the ONE ``[H, T, S]`` slab and ``get_attn_gradients()`` the ``[batch*H, T, S]`` slab ``backward_shared`` fills.
        ``key is query`` (DETR's self-attentions: q = k = x + pos): q and k come out of ONE GEMM against the packed
        ``[Wq; Wk]``.  ``kv = (k, v)`` ``[1, S, H, D]`` views: projections the caller already has (the decoder's
        cross-attentions read the same memory in every layer: all layers' keys / values are two GEMMs)."""
        T, H, D = query.shape[1], self.num_heads, self.head_dim
        if kv is not None:
            q = ops.small_linear(query, self.q_proj).view(1, T, H, D)
            k, v = kv
        elif key is query:
            from .bert_tape import packed_linear
            W, b = packed_linear((self.q_proj, self.k_proj))
            if self.overlap_value_proj:
                # the value projection beside the q / k one: at one sample these are launch-latency-sized GEMMs
                main, side = torch.cuda.current_stream(), ops.side_stream(query.device)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    v = ops.small_linear(value, self.v_proj).view(1, -1, H, D)
                qk = torch.addmm(b, query.reshape(T, -1), W.t()).view(1, T, 2, H, D)
                main.wait_stream(side)
                v.record_stream(main)
            else:
                v = ops.small_linear(value, self.v_proj).view(1, -1, H, D)
                qk = torch.addmm(b, query.reshape(T, -1), W.t()).view(1, T, 2, H, D)
            q, k = qk[:, :, 0], qk[:, :, 1]
        else:
            q = self.q_proj(query).view(1, T, H, D)
            k = self.k_proj(key).view(1, -1, H, D)
            v = self.v_proj(value).view(1, -1, H, D)
        S = k.shape[1]
        dev = query.device
        if self._probs is None or tuple(self._probs.shape) != (1, H, T, S) or self._probs.device != dev:
            self._probs = torch.empty(1, H, T, S, dtype=torch.float32, device=dev)
        if self._grads is None or tuple(self._grads.shape) != (batch, H, T, S) or self._grads.device != dev:
            self._grads = torch.empty(batch, H, T, S, dtype=torch.float32, device=dev)
        o = ops.attn_capture_fwd(q, k, v, self._probs, float(D) ** -0.5, _lib.SCALE_Q_FIRST, None, layout="bnhd")
        self.save_attn(self._probs.view(H, T, S))
        self.save_attn_gradients(self._grads.view(batch * H, T, S))
        return ops.small_linear(o.reshape(1, T, self.embed_dim), self.out_proj), (q, k, v, o)

    @torch.no_grad()
    def backward_shared(self, tape, d_out, need_input_grads=True, gemm_dtype=torch.float32):
        """``d_out [K, T, E]`` -> ``(d_query [K, T, E], d_key [K, S, E], d_value [K, S, E])`` (``None`` when not needed);
        always writes dL/dP of the K samples into the gradient slab."""
        q, k, v, o_fwd = tape
        K, T = d_out.shape[0], d_out.shape[1]
        d_o = ops.backward_gemm(d_out, self.out_proj.weight, gemm_dtype).view(K, T, self.num_heads, self.head_dim)
        dq, dk, dv = ops.attn_capture_bwd(q, k, v, self._probs, d_o, self._grads, float(self.head_dim) ** -0.5,
                                          _lib.SCALE_Q_FIRST, need_dqkv=need_input_grads, layout="bnhd", batch=K,
                                          o=o_fwd)
        if not need_input_grads:
            return None, None, None
        E = self.embed_dim
        return (ops.backward_gemm(dq.reshape(K, T, E), self.q_proj.weight, gemm_dtype),
                ops.backward_gemm(dk.reshape(K, -1, E), self.k_proj.weight, gemm_dtype),
                ops.backward_gemm(dv.reshape(K, -1, E), self.v_proj.weight, gemm_dtype))

    @torch.no_grad()
    def backward_shared_into(self, tape, d_out, acc_q, acc_kv=None, same_source=True, kv_weights=None):
        """``backward_shared`` with the input-gradient GEMMs ACCUMULATING (beta = 1) into buffers the caller owns, fp32.

        ``same_source=True`` (self-attention whose q, k and v inputs are the same tensor up to the constant positional term):
        the attention backward writes one packed ``[K, T, 3, H, D]`` gradient and ``acc_q [K, T, E] += [dq | dk | dv] .
        [Wq; Wk; Wv]`` is ONE GEMM (instead of three GEMMs and three adds).  Returns ``acc_q``.
        ``same_source=False`` (cross-attention): ``acc_q += dq . Wq``; ``acc_kv [K, S, E] += [dk | dv] . [Wk; Wv]`` (``acc_kv=None``:
        a fresh product).  ``kv_weights``: the packed ``[2E, E]`` weight when the caller holds it.  Returns ``(acc_q, acc_kv)``."""
        from .bert_tape import packed_linear
        q, k, v, o_fwd = tape
        K, T, E = d_out.shape[0], d_out.shape[1], self.embed_dim
        H, D = self.num_heads, self.head_dim
        d_o = torch.matmul(d_out, self.out_proj.weight).view(K, T, H, D)
        scale = float(D) ** -0.5
        if same_source:
            dqkv = torch.empty(K, T, 3, H, D, dtype=torch.float32, device=d_out.device)
            ops.attn_capture_bwd(q, k, v, self._probs, d_o, self._grads, scale, _lib.SCALE_Q_FIRST, need_dqkv=True,
                                 layout="bnhd", batch=K, o=o_fwd, out=(dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2]))
            W, _ = packed_linear((self.q_proj, self.k_proj, self.v_proj))
            acc_q.view(K * T, E).addmm_(dqkv.view(K * T, 3 * E), W)
            return acc_q
        S = k.shape[1]
        dq = torch.empty(K, T, H, D, dtype=torch.float32, device=d_out.device)
        dkv = torch.empty(K, S, 2, H, D, dtype=torch.float32, device=d_out.device)
        ops.attn_capture_bwd(q, k, v, self._probs, d_o, self._grads, scale, _lib.SCALE_Q_FIRST, need_dqkv=True,
                             layout="bnhd", batch=K, o=o_fwd, out=(dq, dkv[:, :, 0], dkv[:, :, 1]))
        acc_q.view(K * T, E).addmm_(dq.view(K * T, E), self.q_proj.weight)
        W = kv_weights if kv_weights is not None else packed_linear((self.k_proj, self.v_proj))[0]
        flat = dkv.view(K * S, 2 * E)
        acc_kv = torch.mm(flat, W).view(K, S, E) if acc_kv is None else acc_kv.view(K * S, E).addmm_(flat, W).view(K, S, E)
        return acc_q, acc_kv


class BertStyleAttention(_SlabOwner):
    """LXMERT ``LxmertAttention`` / BERT ``BertSelfAttention`` on the HIP capture op (see module docstring)."""

    def __init__(self, hidden_size, num_attention_heads, ctx_dim=None, share_weights_with=None):
        super().__init__()
        if hidden_size % num_attention_heads:
            raise ValueError("hidden size must be a multiple of the number of heads")
        self.num_attention_heads = num_attention_heads
        self.attention_head_size = hidden_size // num_attention_heads
        self.head_size = hidden_size
        ctx_dim = hidden_size if ctx_dim is None else ctx_dim
        if share_weights_with is not None:
            # LXMERT runs its cross-attention weights twice per x-layer (text->image and image->text,
            # lxmert_lrp.py:640-656 deep-copies the module so that each direction keeps its own saved attention);
            # here the second direction shares the projection modules and owns only its capture slabs
            self.query, self.key, self.value = (share_weights_with.query, share_weights_with.key,
                                                share_weights_with.value)
            return
        self.query = nn.Linear(hidden_size, hidden_size)
        self.key = nn.Linear(ctx_dim, hidden_size)
        self.value = nn.Linear(ctx_dim, hidden_size)

    def forward(self, hidden_states, context=None, attention_mask=None, output_attentions=False):
        self._lrp_tape = None            # (also on the empty-stream early return and under no_grad: nothing stale stays pinned)
        context = hidden_states if context is None else context
        B, Nq, _ = hidden_states.shape
        Nk = context.shape[1]
        H, D = self.num_attention_heads, self.attention_head_size
        q = self.query(hidden_states).view(B, Nq, H, D)
        k = self.key(context).view(B, Nk, H, D)
        v = self.value(context).view(B, Nk, H, D)
        if Nq == 0 or Nk == 0:
            # an empty stream (the perturbation evaluator's "no region kept" step): softmax over no keys contributes
            # nothing, exactly like torch's matmul with an empty inner dimension
            probs = hidden_states.new_zeros(B, H, Nq, Nk)
            self.save_attn(probs)
            self.save_attn_gradients(torch.zeros_like(probs))
            context_layer = hidden_states.new_zeros(B, Nq, H * D)
            return (context_layer, probs) if output_attentions else (context_layer,)
        mask = None
        if attention_mask is not None:   # HF extended mask [B, 1, 1, Nk] (additive, -10000 on padding)
            mask = attention_mask.reshape(B, 1, Nk).float()
        probs, grads = self._slabs(B, H, Nq, Nk, hidden_states.device)
        o = attention_capture(q, k, v, probs, grads, math.sqrt(D), mask=mask, scale_mode=_lib.SCALE_SCORES)
        self.save_attn(probs)
        self.save_attn_gradients(grads)
        if torch.is_grad_enabled():      # what the body's LRP pass reads (bert_lrp.attention_relprop); references, no copies
            self._lrp_tape = dict(hidden=hidden_states.detach(), context=context.detach(), q=q.detach(), k=k.detach(),
                                  v=v.detach(), o=o.detach().view(B, Nq, H, D), probs=probs, mask=attention_mask)
        context_layer = o.reshape(B, Nq, H * D)
        return (context_layer, probs) if output_attentions else (context_layer,)
