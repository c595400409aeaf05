"""Thin torch-tensor front end over the C-ABI (``_lib``): pointer/stride plumbing only.

Every function requires HIP-device tensors and launches on torch's current stream.  There is no CPU
path and no PyTorch-op fallback: a CPU tensor or a missing ``libmmx_hip.so`` raises.
"""
from __future__ import annotations

import contextlib
import threading
import ctypes as C
import gc
import os
import weakref

import torch

from . import _lib
from ._lib import MMXError, check, lib

_DTYPES = {torch.float32: _lib.MMX_F32, torch.float16: _lib.MMX_F16, torch.bfloat16: _lib.MMX_BF16}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise MMXError("mmx ops need HIP-device tensors (got %s); there is no CPU path" % t.device)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _f32c(t):
    """fp32 + contiguous (no copy when already so)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _capture(t):
    if t.dtype not in _DTYPES:
        raise MMXError("captured buffers must be fp32/fp16/bf16, got %s" % t.dtype)
    return t.contiguous()


_ws_cache = {}


def set_option(key, value):
    """Process-wide tuning knob of the C library (see ``mmx_set_option`` in include/mmx_relevancy.h)."""
    check(lib().mmx_set_option(key.encode(), int(value)), "mmx_set_option")


def _workspace(nbytes, device, tag="default"):
    """Grow-only scratch per (device, stream, tag): reuse is stream-ordered, so every stream owns its own."""
    if nbytes == 0:
        return None
    key = (device, torch.cuda.current_stream(device).cuda_stream, tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


_SIDE_STREAMS = {}
_CAPTURE = threading.local()     # .side: the private side stream of the capture in progress on this thread (graph_capture)


def side_stream(device, slot=0):
    """The side stream independent chains of an explainability pass run on beside the current stream (LXMERT's two modalities,
    DETR's / ViT's rule kernels beside the backward, CLIP's second tower).  Eager calls share one stream per (device, slot),
    created once; INSIDE ``graph_capture`` every capture gets a PRIVATE side stream (and a private capture stream) that lives
    and dies with its graph.  Why: graphs captured across streams that earlier, since destroyed, graphs had also been captured
    across segfaulted inside ``hipGraphLaunch`` (round 3: the DETR pass after CLIP / ViT graphs in one process, ``bench.py``'s
    config legs and the GPU test suite) -- with no stream shared between two captures the same sequences are clean.
    Use ONE side stream per pass."""
    if os.environ.get("MMX_DEBUG_NO_SIDE"):          # debugging aid (MMX_DEBUG_NO_SIDE=1): every chain on the current stream
        return torch.cuda.current_stream(device)
    private = getattr(_CAPTURE, "side", None)
    if private is not None:
        return private
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), slot)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=key[0])
    return _SIDE_STREAMS[key]


@contextlib.contextmanager
def graph_capture(graph):
    """``torch.cuda.graph(graph, ...)`` the way every ``Graphed*`` wrapper of this package captures:

    * Python's cyclic garbage collector is held off for the duration of the capture.  A collection that happens to run INSIDE a
      capture may destroy objects whose destructors issue HIP calls that are illegal while a stream is capturing -- a
      ``torch.cuda.CUDAGraph`` of an earlier, already dropped wrapper releases its private memory pool (``hipFree``) -- and an
      exception in a C++ destructor aborts the process (seen as "Fatal Python error: Aborted ... Garbage-collecting" in the GPU
      test suite).  ``torch.cuda.graph`` collects once BEFORE the capture begins; this keeps the collector from firing again.
    * the capture runs on its OWN capture stream and ``side_stream()`` hands out a side stream private to this capture (see
      there); both, and the scratch buffers first allocated on them, are kept on the graph object and released with it.
    * ``capture_error_mode="thread_local"``: a collective backend's watchdog thread (RCCL, one rank per GPU) may poll events
      while this thread captures; only calls made by the capturing thread must be capture-safe."""
    was_enabled = gc.isenabled()
    gc.collect()
    gc.disable()
    capture_stream, private_side = torch.cuda.Stream(), torch.cuda.Stream()
    known = set(_ws_cache)
    _CAPTURE.side = private_side
    try:
        with torch.cuda.graph(graph, stream=capture_stream, capture_error_mode="thread_local"):
            yield
    finally:
        _CAPTURE.side = None
        # scratch first allocated on the private streams belongs to this graph alone: off the shared cache, onto the graph
        graph._mmx_private = ([_ws_cache.pop(k) for k in list(_ws_cache) if k not in known], capture_stream, private_side)
        if was_enabled:
            gc.enable()


def pinned_state(model=None):
    """Everything a captured hipGraph may have baked raw addresses of, besides its own pool: the grow-only scratch
    buffers of this module and (``model`` given) the capture slabs of every hooked attention module.  A ``Graphed*``
    wrapper keeps the returned list alive: a later EAGER call that needs a bigger scratch buffer or other slab shapes
    replaces those tensors, and without this reference the old ones would be freed under the graph."""
    keep = list(_ws_cache.values())
    if model is not None:
        for mod in model.modules():
            for name in ("_probs", "_grads"):
                t = getattr(mod, name, None)
                if torch.is_tensor(t):
                    keep.append(t)
            buf = getattr(mod, "buffers", None)
            if buf is not None and hasattr(buf, "probs") and hasattr(buf, "grads"):
                keep.extend([buf.probs, buf.grads, buf])
    return keep


# ------------------------------------------------------------------------------------------- rule 5
def avg_heads(cam, grad, batch_size=1, shared_attn=False):
    """``mean_h(clamp(grad*cam, 0))`` -> ``[batch_size, Nq, Nk]`` fp32 (leading dims flattened into B*H).
    ``shared_attn=True``: ``cam`` holds ONE sample's heads, shared by the ``batch_size`` samples of ``grad``."""
    _dev(cam, grad)
    cam, grad = _capture(cam), _capture(grad)
    nq, nk = grad.shape[-2], grad.shape[-1]
    bh = grad.numel() // (nq * nk)
    ok = cam.numel() * batch_size == grad.numel() if shared_attn else cam.shape == grad.shape
    if not ok or cam.shape[-2:] != grad.shape[-2:] or cam.dtype != grad.dtype:
        raise MMXError("avg_heads: cam %s/%s vs grad %s/%s" % (tuple(cam.shape), cam.dtype, tuple(grad.shape), grad.dtype))
    if bh % batch_size:
        raise MMXError("avg_heads: %d matrices not divisible by batch %d" % (bh, batch_size))
    out = torch.empty(batch_size, nq, nk, dtype=torch.float32, device=cam.device)
    check(lib().mmx_avg_heads_ex(_p(cam), _p(grad), _p(out), batch_size, bh // batch_size, nq, nk, _DTYPES[cam.dtype],
                                 0 if shared_attn else -1, _stream()), "mmx_avg_heads_ex")
    return out


# ------------------------------------------------------------------------------------------- fused chain
def relevancy_self_chain(attn_layers, grad_layers, batch_size, R_init=None, R_sq_init=None, shared_attn=False, causal=False):
    """All-layer chain ``R <- R + A_bar_l @ R`` (``R_0 = I`` or ``R_init``) [and ``R_sq`` likewise] in one call.

    ``attn_layers[l]`` / ``grad_layers[l]``: ``[B*H, N, N]`` or ``[B, H, N, N]`` (fp32/fp16/bf16).
    ``shared_attn=True``: ``attn_layers[l]`` holds ONE sample's heads (``[H, N, N]``) shared by all ``batch_size``
    samples of ``grad_layers[l]`` (shared-forward mode; read with batch stride 0).
    ``causal=True``: the probabilities come from causally masked attention (exact zeros above the diagonal, CLIP's text tower):
    the kernels do not read what lies above the diagonal (``MMX_CHAIN_CAUSAL``); same bits, about half the bytes.
    Returns ``R [B, N, N]`` (and ``R_sq [B, N, M]`` when ``R_sq_init`` is given).
    """
    if len(attn_layers) != len(grad_layers):
        raise MMXError("self_chain: %d attn layers vs %d grad layers" % (len(attn_layers), len(grad_layers)))
    if len(attn_layers) > _lib.MAX_LAYERS:
        raise MMXError("self_chain: more than %d layers" % _lib.MAX_LAYERS)
    _dev(*attn_layers, *grad_layers, R_init, R_sq_init)
    attn = [_capture(a) for a in attn_layers]
    grad = [_capture(g) for g in grad_layers]
    ref = attn[0] if attn else R_init
    if ref is None:
        raise MMXError("self_chain: no layers and no R_init")
    n = ref.shape[-1]
    device = ref.device
    dt = _DTYPES[attn[0].dtype] if attn else _lib.MMX_F32
    heads = 1
    for a, g in zip(attn, grad):
        if (a.shape != g.shape and not shared_attn) or a.dtype != attn[0].dtype or g.dtype != attn[0].dtype \
                or a.shape[-1] != n or a.shape[-2] != n or (shared_attn and a.numel() * batch_size != g.numel()):
            raise MMXError("self_chain: inconsistent layer shapes/dtypes")
    if attn:
        bh = grad[0].numel() // (n * n)
        if bh % batch_size:
            raise MMXError("self_chain: %d matrices not divisible by batch %d" % (bh, batch_size))
        heads = bh // batch_size
    R_out = torch.empty(batch_size, n, n, dtype=torch.float32, device=device)
    if R_init is not None:
        R_init = _f32c(R_init).expand(batch_size, n, n).contiguous()
    m = 0
    sq_out = None
    if R_sq_init is not None:
        R_sq_init = _f32c(R_sq_init)
        m = R_sq_init.shape[-1]
        R_sq_init = R_sq_init.expand(batch_size, n, m).contiguous()
        sq_out = torch.empty(batch_size, n, m, dtype=torch.float32, device=device)
    need = lib().mmx_self_chain_workspace_bytes(len(attn), batch_size, heads, n, m, dt)
    ws = _workspace(need, device)
    at, _k1 = _lib.ptr_table([a.data_ptr() for a in attn])
    gt, _k2 = _lib.ptr_table([g.data_ptr() for g in grad])
    check(lib().mmx_relevancy_self_chain_flags(at, gt, len(attn), batch_size, heads, n, dt, 0 if shared_attn else -1,
                                               _p(R_init), _p(R_out), _p(R_sq_init), _p(sq_out), m,
                                               _lib.CHAIN_CAUSAL if causal else 0, _p(ws), need,
                                               _stream()), "mmx_relevancy_self_chain_flags")
    return (R_out, sq_out) if sq_out is not None else R_out


def relevancy_chain_row(attn_layers, grad_layers, batch_size, rows, shared_attn=False):
    """Row ``rows[b]`` of sample b's ``relevancy_self_chain`` result, ``[B, N]`` -- what the ViT / VisualBERT generators
    return (``R[0, 1:]``, ViT notebook cell 7:34; ``R[cls_index]``, VisualBERT/.../ExplanationGenerator.py:95-97).

    ``R = (I + A_L) ... (I + A_1)``, so ``e_r^T R`` is a row vector carried from the TOP layer down,
    ``x <- x + x A_l`` (``mmx_chain_vecmat``): per layer one head-average pass over the slabs and one mat-vec, instead of
    the N^3 product the long-sequence split path runs.  N <= 128 keeps the one-launch fused kernel (all layers, R in
    registers) and picks the row afterwards -- that is faster there.  Same result up to fp32 summation order."""
    n = grad_layers[0].shape[-1]
    dev = grad_layers[0].device
    fixed = isinstance(rows, int)               # (a Python int needs no host -> device copy: capturable into a hipGraph)
    if not fixed:
        rows = torch.as_tensor(rows, device=dev).reshape(-1).expand(batch_size)
    if n <= 128:
        R = relevancy_self_chain(attn_layers, grad_layers, batch_size, shared_attn=shared_attn)
        return R[:, rows].contiguous() if fixed else R[torch.arange(batch_size, device=dev), rows]
    x = torch.zeros(batch_size, n, dtype=torch.float32, device=dev)
    if fixed:
        x[:, rows] = 1.0
    else:
        x.scatter_(1, rows.reshape(batch_size, 1), 1.0)
    for a, g in zip(reversed(list(attn_layers)), reversed(list(grad_layers))):
        x = chain_vecmat(x, avg_heads(a, g, batch_size=batch_size, shared_attn=shared_attn))
    return x


class _QuickGELU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y = torch.empty_like(x)
        check(lib().mmx_quick_gelu_fwd(_p(x), _p(y), x.numel(), _stream()), "mmx_quick_gelu_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        check(lib().mmx_quick_gelu_bwd(_p(x), _p(dy), _p(dx), x.numel(), _stream()), "mmx_quick_gelu_bwd")
        return dx


def quick_gelu(x):
    """``x * sigmoid(1.702 x)`` as one fused pass forward and one backward (fp32 HIP tensors)."""
    _dev(x)
    if x.dtype != torch.float32:
        raise MMXError("quick_gelu: fp32 only")
    return _QuickGELU.apply(x)


def quick_gelu_fwd(x, out_dtype=torch.float32):
    """``x * sigmoid(1.702 x)``, no autograd (the hand-written backward passes keep ``x`` themselves).  ``out_dtype=torch.bfloat16``
    (a bf16 body: the activation only feeds the next GEMM): the kernel writes bf16, no conversion pass."""
    _dev(x)
    x = _f32c(x)
    if out_dtype == torch.bfloat16 and x.numel() % 4 == 0:
        y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
        check(lib().mmx_quick_gelu_fwd_bf16(_p(x), _p(y), x.numel(), _stream()), "mmx_quick_gelu_fwd_bf16")
        return y
    y = torch.empty_like(x)
    check(lib().mmx_quick_gelu_fwd(_p(x), _p(y), x.numel(), _stream()), "mmx_quick_gelu_fwd")
    return y


def quick_gelu_bwd(x, dy):
    """``dy * QuickGELU'(x)`` in one pass.  ``x`` has ``dy``'s shape, or batch 1 where ``dy`` has batch B (shared-forward
    mode: the ONE activation tensor is broadcast over the B upstream gradients inside the kernel)."""
    _dev(x, dy)
    if dy.dtype == torch.bfloat16:             # bf16 gradient stream: dy / dx bf16, x fp32 (broadcast over the batch)
        x, dy = _f32c(x), dy.contiguous()
        if dy.numel() % x.numel() or x.numel() % 8:
            raise MMXError("quick_gelu_bwd (bf16): x %s does not tile dy %s" % (tuple(x.shape), tuple(dy.shape)))
        dx = torch.empty_like(dy)
        check(lib().mmx_quick_gelu_bwd_bcast_bf16(_p(x), _p(dy), _p(dx), dy.numel(), x.numel(), _stream()),
              "mmx_quick_gelu_bwd_bcast_bf16")
        return dx
    x, dy = _f32c(x), _f32c(dy)
    dx = torch.empty_like(dy)
    if x.shape == dy.shape:
        check(lib().mmx_quick_gelu_bwd(_p(x), _p(dy), _p(dx), x.numel(), _stream()), "mmx_quick_gelu_bwd")
    elif x.shape[0] == 1 and x.shape[1:] == dy.shape[1:] and x.numel() % 4 == 0:
        check(lib().mmx_quick_gelu_bwd_bcast(_p(x), _p(dy), _p(dx), dy.numel(), x.numel(), _stream()),
              "mmx_quick_gelu_bwd_bcast")
    else:
        raise MMXError("quick_gelu_bwd: x %s vs dy %s" % (tuple(x.shape), tuple(dy.shape)))
    return dx


def add_layernorm(x, y, gamma, beta, eps=1e-5, h_dtype=torch.float32):
    """``s = x + y; h = LayerNorm(s)`` in one pass -> ``(s, h, mean, rstd)`` (``y=None``: ``s`` is ``x`` itself).
    ``mean`` / ``rstd``: ``[rows]`` fp32, what ``layernorm_bwd_add`` takes.  ``h_dtype=torch.bfloat16``: ``h`` leaves the kernel
    as bf16 (a bf16 body feeds it to a half-precision GEMM only)."""
    _dev(x, y, gamma, beta)
    x = _f32c(x)
    E = x.shape[-1]
    rows = x.numel() // E
    if y is not None:
        y = _f32c(y)
        if y.shape != x.shape:
            raise MMXError("add_layernorm: x %s vs y %s" % (tuple(x.shape), tuple(y.shape)))
    s = torch.empty_like(x) if y is not None else x
    h = torch.empty(x.shape, dtype=h_dtype, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(lib().mmx_add_layernorm_fwd_ex(_p(x), _p(y), _p(_f32c(gamma)), _p(_f32c(beta)), _p(s) if y is not None else _p(None),
                                         _p(h), _p(mean), _p(rstd), rows, E, float(eps), _DTYPES[h_dtype], _stream()),
          "mmx_add_layernorm_fwd")
    return s, h, mean, rstd


def layernorm_bwd_add(dy, x, mean, rstd, gamma, d_res=None):
    """``d_res + LayerNorm'(dy)`` with forward statistics shared across the batch: ``dy``/``d_res`` ``[B, N, E]``,
    ``x`` ``[1, N, E]`` (or ``[B, N, E]``), ``mean``/``rstd`` matching ``x``'s rows."""
    _dev(dy, x, mean, rstd, gamma, d_res)
    dy, x = _f32c(dy), _f32c(x)
    E = dy.shape[-1]
    rows, x_rows = dy.numel() // E, x.numel() // E
    dx = torch.empty_like(dy)
    check(lib().mmx_layernorm_bwd_add(_p(dy), _p(x), _p(_f32c(mean)), _p(_f32c(rstd)), _p(_f32c(gamma)),
                                      _p(_f32c(d_res)) if d_res is not None else _p(None), _p(dx), rows, x_rows, E,
                                      _stream()), "mmx_layernorm_bwd_add")
    return dx


def layernorm_bwd_add_bf16(dy, x, mean, rstd, gamma, d_res=None, want_f32=True):
    """``layernorm_bwd_add`` for a bf16 upstream gradient ``dy``: returns ``(dx fp32 or None, dx bf16)`` -- the fp32 result
    is the next residual gradient, the bf16 copy the operand of the next bf16 GEMM (one pass writes both)."""
    _dev(dy, x, mean, rstd, gamma, d_res)
    if dy.dtype != torch.bfloat16:
        raise MMXError("layernorm_bwd_add_bf16: dy must be bf16")
    dy, x = dy.contiguous(), _f32c(x)
    E = dy.shape[-1]
    rows, x_rows = dy.numel() // E, x.numel() // E
    dx = torch.empty(dy.shape, dtype=torch.float32, device=dy.device) if want_f32 else None
    dx_h = torch.empty_like(dy)
    check(lib().mmx_layernorm_bwd_add_bf16(_p(dy), _p(x), _p(_f32c(mean)), _p(_f32c(rstd)), _p(_f32c(gamma)),
                                           _p(_f32c(d_res)) if d_res is not None else _p(None), _p(dx), _p(dx_h), rows,
                                           x_rows, E, _stream()), "mmx_layernorm_bwd_add_bf16")
    return dx, dx_h


_GEMM_WEIGHTS = {}   # id(parameter) -> {dtype: (version, converted copy)}; the entry is dropped when the parameter dies


def _converted(weight, dtype):
    per_weight = _GEMM_WEIGHTS.get(id(weight))
    if per_weight is None:
        per_weight = _GEMM_WEIGHTS[id(weight)] = {}
        weakref.finalize(weight, _GEMM_WEIGHTS.pop, id(weight), None)
    hit = per_weight.get(dtype)
    if hit is None or hit[0] != weight._version or hit[1].device != weight.device:
        hit = per_weight[dtype] = (weight._version, weight.detach().to(dtype))
    return hit[1]


def transposed_weight(weight):
    """Cached contiguous ``weight.t()`` (until the parameter is modified in place).  The input-gradient product ``dY . W`` of
    a Linear with few outputs and many inputs (DETR's ``linear2``: 256 -> 2048 on the way back) gets a poor library kernel in its
    natural NN layout (64x64x16 tiles: 140 us at 9500 rows); as ``F.linear(dY, W^T)`` the heuristic picks a 256-wide one
    (104 us) -- profiles/r03_detr_probe.txt."""
    per_weight = _GEMM_WEIGHTS.get(id(weight))
    if per_weight is None:
        per_weight = _GEMM_WEIGHTS[id(weight)] = {}
        weakref.finalize(weight, _GEMM_WEIGHTS.pop, id(weight), None)
    hit = per_weight.get("t")
    if hit is None or hit[0] != weight._version or hit[1].device != weight.device:
        hit = per_weight["t"] = (weight._version, weight.detach().t().contiguous())
    return hit[1]


def sign_split_weight(weight, transposed=False):
    """Cached ``[clamp(W, min=0) | clamp(W, max=0)]`` (``[out, 2 in]``, until the parameter is modified in place): the weight
    operand of both GEMMs of the fused ``Linear.relprop`` (``lrp_linear``); ``transposed``: its contiguous ``[2 in, out]`` form."""
    per_weight = _GEMM_WEIGHTS.get(id(weight))
    if per_weight is None:
        per_weight = _GEMM_WEIGHTS[id(weight)] = {}
        weakref.finalize(weight, _GEMM_WEIGHTS.pop, id(weight), None)
    hit = per_weight.get("pn")
    if hit is None or hit[0] != weight._version or hit[1].device != weight.device:
        w = weight.detach().float()
        hit = per_weight["pn"] = (weight._version, torch.cat((w.clamp(min=0), w.clamp(max=0)), dim=1).contiguous())
        per_weight.pop("pnT", None)
    if not transposed:
        return hit[1]
    hit_t = per_weight.get("pnT")
    if hit_t is None or hit_t[0] != weight._version or hit_t[1].device != weight.device:
        hit_t = per_weight["pnT"] = (weight._version, hit[1].t().contiguous())
    return hit_t[1]


def lrp_fusable(*tensors):
    """The fused LRP rule kernels take fp32 CUDA tensors (anything else -- the CPU suite's host-logic tests -- runs the torch
    formulation in ``lrp.py``)."""
    return all(t.is_cuda and t.dtype == torch.float32 for t in tensors)


def lrp_linear(R, X, weight, normalize):
    """``Linear.relprop`` (alpha = 1) as 2 GEMMs + 3 (4 with ``normalize``) launches: see ``csrc/lrp_kernels.hip``.
    ``R [..., out]``, ``X [..., in]``, ``weight [out, in]`` -> ``[..., in]``."""
    n_in, n_out = X.shape[-1], R.shape[-1]
    # the kernels index R by X's rows and the weight by (R, X)'s last dimensions: a mismatched or broadcastable R must not reach them
    if tuple(R.shape[:-1]) != tuple(X.shape[:-1]) or tuple(weight.shape) != (n_out, n_in):
        raise MMXError("lrp_linear: R %s, X %s, weight %s do not describe Linear.relprop (R [..., out], X [..., in], weight "
                       "[out, in] with equal leading dimensions)" % (tuple(R.shape), tuple(X.shape), tuple(weight.shape)))
    Xc, Rc = X.contiguous(), R.contiguous()
    rows = Xc.numel() // n_in
    WW = sign_split_weight(weight)                                   # [out, 2 in]
    st = _stream()
    XX = torch.empty(rows, 2 * n_in, dtype=torch.float32, device=X.device)
    check(lib().mmx_lrp_split_signs(_p(Xc), _p(XX), rows, n_in, st), "mmx_lrp_split_signs")

    def mm(a, wt_of, k, n):       # a [rows, k] . Wt [k, n]; few rows and a short contraction: the library's heuristic picks a
        if rows <= 128 and k <= 1024:   # 256-row tile on ONE workgroup (30-40 us, r04_lrp_probe.txt) -> our 32 x 32-tile kernel
            out_ = torch.empty(rows, n, dtype=torch.float32, device=a.device)
            check(lib().mmx_linear_f32(_p(a), _p(wt_of()), None, _p(out_), rows, n, k, st), "mmx_linear_f32")
            return out_
        return torch.mm(a, wt_of())
    Z = mm(XX, lambda: sign_split_weight(weight, transposed=True) if rows <= 128 and 2 * n_in <= 1024 else WW.t(), 2 * n_in, n_out)
    S = torch.empty_like(Z)                                          # Z [rows, out] = px pw^T + nx nw^T  (K = 2 in)
    check(lib().mmx_lrp_safe_divide(_p(Rc), _p(Z), _p(S), Z.numel(), st), "mmx_lrp_safe_divide")
    Y = mm(S, lambda: WW, n_out, 2 * n_in)                           # [rows, 2 in] = [S pw | S nw]
    out = torch.empty(X.shape, dtype=torch.float32, device=X.device)
    ws = _workspace(lib().mmx_lrp_workspace_bytes(), X.device, tag="lrp") if normalize else None
    check(lib().mmx_lrp_linear_combine(_p(XX), _p(Y), _p(out), rows, n_in, _p(Rc) if normalize else None, Rc.numel(),
                                       _p(ws), st), "mmx_lrp_linear_combine")
    return out


def lrp_mha_rescale(cam_v_pre, cam_v_post, cam_k, cam_q, cam_o):
    """The q / k rescale branch of ``MultiheadAttention.relprop`` (layers.py:791-799) in two launches; returns ``(cam_k, cam_q)``
    (fresh contiguous tensors, scaled in place when the branch applies)."""
    vp, vq = cam_v_pre.contiguous(), cam_v_post.contiguous()
    ck, cq, co = cam_k.contiguous().clone(), cam_q.contiguous().clone(), cam_o.contiguous()
    ws = _workspace(lib().mmx_lrp_workspace_bytes(), cam_k.device, tag="lrp")
    check(lib().mmx_lrp_mha_rescale(_p(vp), vp.numel(), _p(vq), vq.numel(), _p(ck), ck.numel(), _p(cq), cq.numel(), _p(co),
                                    co.numel(), _p(ws), _stream()), "mmx_lrp_mha_rescale")
    return ck, cq


def lrp_add(R, a, b, per_sample):
    """``Add.relprop`` in two launches -> ``(R_a, R_b)``; ``per_sample``: the rule's three sums per leading-dimension item."""
    Rc, ac, bc = R.contiguous(), a.contiguous(), b.contiguous()
    batch = Rc.shape[0] if (per_sample and Rc.dim() > 1) else 1
    if batch > 64:
        return None
    ra, rb = torch.empty_like(Rc), torch.empty_like(Rc)
    ws = _workspace(lib().mmx_lrp_workspace_bytes(), R.device, tag="lrp")
    check(lib().mmx_lrp_add_relprop(_p(Rc), _p(ac), _p(bc), _p(ra), _p(rb), batch, Rc.numel() // batch, _p(ws), _stream()),
          "mmx_lrp_add_relprop")
    return ra, rb


def lrp_clone(Rs, X):
    """``Clone.relprop`` in one launch (up to 8 relevance tensors)."""
    Xc = X.contiguous()
    rc = [r.contiguous() for r in Rs]
    out = torch.empty_like(Xc)
    tbl, _k = _lib.ptr_table([r.data_ptr() for r in rc])
    check(lib().mmx_lrp_clone_relprop(tbl, len(rc), _p(Xc), _p(out), Xc.numel(), _stream()), "mmx_lrp_clone_relprop")
    return out


_MM_OUT_DTYPE = [None]      # does torch.mm(a, b, out_dtype=torch.float32) work on this build / device?  probed once


def _mm_f32_out(a, b):
    """``a @ b`` for 2-D half-precision operands with an fp32 result straight from the GEMM's fp32 accumulators
    (``aten::mm.dtype``) -- no rounding of the result to bf16 and no conversion pass; falls back to ``.float()``."""
    if _MM_OUT_DTYPE[0] is None:
        try:
            torch.mm(a[:1], b, out_dtype=torch.float32)
            _MM_OUT_DTYPE[0] = True
        except (RuntimeError, TypeError):
            _MM_OUT_DTYPE[0] = False
    if _MM_OUT_DTYPE[0]:
        return torch.mm(a, b, out_dtype=torch.float32)
    return torch.mm(a, b).float()


def backward_gemm_bf16(x, weight):
    """``x (bf16) @ weight`` -> bf16: a GEMM of the bf16 gradient stream (fp32 accumulation inside the library kernel,
    result rounded once); the bf16 copy of ``weight`` is cached like ``backward_gemm``'s."""
    w = _converted(weight, torch.bfloat16)
    return torch.mm(x.reshape(-1, x.shape[-1]), w).view(*x.shape[:-1], w.shape[-1])


def backward_gemm(x, weight, dtype=torch.float32):
    """``x @ weight`` for the hand-written (shared-forward) backward passes.  ``dtype=torch.bfloat16`` (opt-in per tower /
    body, ``backward_gemm_dtype``) runs it on the bf16 MFMA with fp32 accumulation and an fp32 result; the converted
    weight is cached until the parameter is modified in place.  A plain library GEMM either way (hipBLASLt / rocBLAS)."""
    if dtype == torch.float32:
        return torch.matmul(x, weight)
    w = _converted(weight, dtype)
    return _mm_f32_out(x.reshape(-1, x.shape[-1]).to(dtype), w).view(*x.shape[:-1], w.shape[-1])


def linear(x, weight, bias=None, dtype=torch.float32):
    """``F.linear`` for the tape forward passes; ``dtype=torch.bfloat16`` (a bf16 body, ``forward_gemm_dtype``) rounds
    ``x`` and the (cached) weight to bf16, accumulates in fp32 and returns fp32."""
    if dtype == torch.float32:
        return torch.nn.functional.linear(x, weight, bias)
    w = _converted(weight, dtype)
    y = _mm_f32_out(x.reshape(-1, x.shape[-1]).to(dtype), w.t()).view(*x.shape[:-1], w.shape[0])
    return y if bias is None else y.add_(bias)


class ChainPlan:
    """A prepared ``relevancy_self_chain`` launch over persistent slabs: pointer tables, scratch and the output tensor are
    built once; ``launch()`` is a single C call (no per-call Python tensor plumbing).  The captured slabs of a tower
    keep their addresses from step to step, so the plan stays valid as long as the tensors it references are alive."""

    def __init__(self, attn_layers, grad_layers, batch_size, shared_attn=False, half_chain=False, causal=False):
        """``half_chain``: the reference's fp16 chain (``mmx_relevancy_self_chain_half``: every tensor-level result of the rule
        rounded to fp16, like the notebook's R on a model after ``convert_weights``); ``launch`` then returns an fp16 tensor."""
        _dev(*attn_layers, *grad_layers)
        self.half_chain = bool(half_chain)
        self.flags = _lib.CHAIN_CAUSAL if causal else 0     # ``causal``: see ``relevancy_self_chain``
        self.attn = [_capture(a) for a in attn_layers]
        self.grad = [_capture(g) for g in grad_layers]
        if not self.attn or len(self.attn) != len(self.grad) or len(self.attn) > _lib.MAX_LAYERS:
            raise MMXError("ChainPlan: need 1..%d (attn, grad) layer pairs" % _lib.MAX_LAYERS)
        g0 = self.grad[0]
        self.n = g0.shape[-1]
        self.batch = batch_size
        self.heads = g0.numel() // (self.n * self.n) // batch_size
        self.dt = _DTYPES[g0.dtype]
        self.shared = 0 if shared_attn else -1
        for a, g in zip(self.attn, self.grad):
            if g.shape != g0.shape or g.dtype != g0.dtype or a.dtype != g0.dtype or \
                    a.numel() * (batch_size if shared_attn else 1) != g.numel():
                raise MMXError("ChainPlan: inconsistent layer shapes/dtypes")
        self.device = g0.device
        if self.half_chain and g0.dtype == torch.bfloat16:
            raise MMXError("ChainPlan: the fp16 chain reads fp32 or fp16 slabs")
        self.need = lib().mmx_self_chain_workspace_bytes(len(self.attn), batch_size, self.heads, self.n, 0, self.dt)
        if self.half_chain and self.n <= 128:
            self.need = 0                                  # one fused launch per call, no scratch
        self.ws = torch.empty(max(self.need, 1), dtype=torch.uint8, device=g0.device)
        self.at, self._k1 = _lib.ptr_table([a.data_ptr() for a in self.attn])
        self.gt, self._k2 = _lib.ptr_table([g.data_ptr() for g in self.grad])

    def launch(self, out=None):
        """Enqueue on torch's current stream; returns ``R [B, N, N]`` (a fresh tensor unless ``out`` is given)."""
        if out is None:
            out = torch.empty(self.batch, self.n, self.n, dtype=torch.float32, device=self.device)
        if self.half_chain:
            check(lib().mmx_relevancy_self_chain_half(self.at, self.gt, len(self.attn), self.batch, self.heads, self.n, self.dt,
                                                      self.shared, _p(out), _p(self.ws), self.need, _stream()),
                  "mmx_relevancy_self_chain_half")
            return out.to(torch.float16)                   # lossless: the kernel rounds every result to fp16
        check(lib().mmx_relevancy_self_chain_flags(self.at, self.gt, len(self.attn), self.batch, self.heads, self.n, self.dt,
                                                   self.shared, None, _p(out), None, None, 0, self.flags, _p(self.ws), self.need,
                                                   _stream()), "mmx_relevancy_self_chain_flags")
        return out


# ------------------------------------------------------------------------------------------- matmul
def matmul(a, b, add_to=None, trans_a=False, nan_to_zero=False):
    """fp32 ``op(a) @ b (+ add_to)`` on the exact-fp32 MFMA; 2-D or batched 3-D (2-D operands broadcast)."""
    _dev(a, b, add_to)
    a, b = _f32c(a), _f32c(b)
    batch = 1
    for t in (a, b):
        if t.dim() == 3:
            batch = max(batch, t.shape[0])
    am, ak = (a.shape[-1], a.shape[-2]) if trans_a else (a.shape[-2], a.shape[-1])
    if b.shape[-2] != ak:
        raise MMXError("matmul: inner dims %d vs %d" % (ak, b.shape[-2]))
    n = b.shape[-1]
    sa = a.shape[-2] * a.shape[-1] if (a.dim() == 3 and a.shape[0] > 1) else 0
    sb = ak * n if (b.dim() == 3 and b.shape[0] > 1) else 0
    squeeze = a.dim() == 2 and b.dim() == 2
    out = torch.empty(batch, am, n, dtype=torch.float32, device=a.device)
    cin = None
    if add_to is not None:
        cin = _f32c(add_to).expand(batch, am, n).contiguous() if add_to.dim() == 2 or add_to.shape[0] != batch \
            else _f32c(add_to)
    check(lib().mmx_bmm_f32(_p(a), _p(b), _p(cin), _p(out), batch, am, n, ak, int(trans_a), sa, sb, am * n,
                            int(nan_to_zero), _stream()), "mmx_bmm_f32")
    return out[0] if squeeze else out


def rows_to_dense(vals, rows, n_tokens):
    """``out [B, n_tokens, E]`` = zeros with ``out[b, rows[b]] = vals[b]`` in ONE launch (``mmx_rows_to_dense``)."""
    _dev(vals, rows)
    vals, rows = _f32c(vals), rows.to(torch.long).contiguous()
    if vals.dim() != 2 or rows.numel() != vals.shape[0]:
        raise MMXError("rows_to_dense: vals [B, E] and rows [B] expected, got %s / %s" % (tuple(vals.shape), tuple(rows.shape)))
    B, E = vals.shape
    out = torch.empty(B, n_tokens, E, dtype=torch.float32, device=vals.device)
    check(lib().mmx_rows_to_dense(_p(vals), _p(rows), _p(out), B, n_tokens, E, _stream()), "mmx_rows_to_dense")
    return out


def rows_add_(dense, rows, vals):
    """``dense[b, rows[b]] += vals[b]`` in place (``mmx_rows_add``); ``dense [B, N, E]`` fp32 contiguous."""
    _dev(dense, rows, vals)
    if dense.dtype != torch.float32 or not dense.is_contiguous():
        raise MMXError("rows_add_: dense must be fp32 contiguous")
    B, N, E = dense.shape
    if tuple(vals.shape) != (B, E) or rows.numel() != B:
        raise MMXError("rows_add_: dense [B, N, E] needs vals [B, E] and rows [B], got %s / %s / %s"
                       % (tuple(dense.shape), tuple(vals.shape), tuple(rows.shape)))
    check(lib().mmx_rows_add(_p(dense), _p(rows.to(torch.long).contiguous()), _p(_f32c(vals)), B, N, E, _stream()), "mmx_rows_add")
    return dense


def small_linear(x, lin):
    """``lin(x)`` for an ``nn.Linear`` applied to a FEW rows (a shared forward's 100 decoder queries) with a SQUARE weight of at
    most 256: the one shape family where the library's heuristic leaves the chip idle (a 256-row tile for 100 rows: 3
    workgroups, 30 us; ``mmx_linear_f32``: 9 us).  Anything else goes to ``lin`` itself -- also an ``x`` that carries an
    autograd graph (the kernel records none: this is an op of the tape passes, which differentiate by hand)."""
    rows = x.numel() // x.shape[-1]
    if not (x.is_cuda and x.dtype == torch.float32 and rows <= 128 and lin.in_features == lin.out_features <= 256) or \
            (torch.is_grad_enabled() and x.requires_grad):
        return lin(x)
    x2 = _f32c(x).view(rows, lin.in_features)
    out = torch.empty(rows, lin.out_features, dtype=torch.float32, device=x.device)
    check(lib().mmx_linear_f32(_p(x2), _p(transposed_weight(lin.weight)), _p(lin.bias.detach() if lin.bias is not None else None),
                               _p(out), rows, lin.out_features, lin.in_features, _stream()), "mmx_linear_f32")
    return out.view(*x.shape[:-1], lin.out_features)


def chain_matvec(A, y, base=None):
    """``base + A @ y`` for ``A [B, N, N]``, ``y`` / ``base [B, N]`` (fp32; ``base`` defaults to ``y``): one column of
    the chain ``R <- R + A R``."""
    _dev(A, y, base)
    A, y = _f32c(A), _f32c(y)
    base = y if base is None else _f32c(base)
    B, N = y.shape
    out = torch.empty_like(y)
    check(lib().mmx_chain_matvec(_p(A), _p(y), _p(base), _p(out), B, N, _stream()), "mmx_chain_matvec")
    return out


def chain_vecmat(x, A, base=None):
    """``base + x @ A`` for ``x`` / ``base [B, N]``, ``A [B, N, N]`` (fp32; ``base`` defaults to ``x``): one row of the
    chain carried top-down."""
    _dev(A, x, base)
    A, x = _f32c(A), _f32c(x)
    base = x if base is None else _f32c(base)
    B, N = x.shape
    out = torch.empty_like(x)
    need = lib().mmx_chain_vecmat_workspace_bytes(B, N)
    ws = _workspace(need, x.device, "vecmat")
    check(lib().mmx_chain_vecmat(_p(A), _p(x), _p(base), _p(out), B, N, _p(ws), need, _stream()), "mmx_chain_vecmat")
    return out


def avg_heads_vecmat(x, cam, grad, batch_size=1, shared_attn=False, base=None):
    """``base + x @ mean_h(clamp(grad * cam, 0))`` for ``x`` / ``base [B, N]`` fp32 (``base`` defaults to ``x``) and square slabs
    ``[B * H, N, N]``: ``chain_vecmat(x, avg_heads(cam, grad))`` without the ``[B, N, N]`` intermediate, two launches instead of
    three (``mmx_avg_heads_vecmat``).  ``shared_attn`` as in ``avg_heads``."""
    _dev(cam, grad, x, base)
    cam, grad, x = _capture(cam), _capture(grad), _f32c(x)
    base = x if base is None else _f32c(base)
    B, N = x.shape
    bh = grad.numel() // (N * N)
    ok = cam.numel() * batch_size == grad.numel() if shared_attn else cam.shape == grad.shape
    if not ok or B != batch_size or bh % B or tuple(grad.shape[-2:]) != (N, N) or cam.dtype != grad.dtype:
        raise MMXError("avg_heads_vecmat: x %s, cam %s/%s, grad %s/%s, batch %d" % (tuple(x.shape), tuple(cam.shape), cam.dtype,
                                                                                     tuple(grad.shape), grad.dtype, batch_size))
    out = torch.empty_like(x)
    need = lib().mmx_avg_heads_vecmat_workspace_bytes(B, N)
    ws = _workspace(need, x.device, "ahv")
    check(lib().mmx_avg_heads_vecmat(_p(cam), _p(grad), _p(x), _p(base), _p(out), B, bh // B, N, _DTYPES[cam.dtype],
                                     0 if shared_attn else -1, _p(ws), need, _stream()), "mmx_avg_heads_vecmat")
    return out


# ------------------------------------------------------------------------------------------- eq. 8-9
def handle_residual(R, check_diag=True):
    """Eq. 8-9 on ``[.., N, N]``.  ``check_diag``: ``True`` asserts ``min diag(R - I) >= 0`` now, like the reference's
    assert (one device->host read); ``False`` skips the check; ``"defer"`` returns ``(out, diag_min)`` with the device
    word, so that a caller can check many calls with ONE read (or none, inside a hipGraph)."""
    _dev(R)
    R = _f32c(R)
    n = R.shape[-1]
    batch = R.numel() // (n * n)
    out = torch.empty_like(R)
    dmin = torch.empty(1, dtype=torch.float32, device=R.device) if check_diag else None
    check(lib().mmx_handle_residual(_p(R), _p(out), batch, n, _p(dmin), _stream()), "mmx_handle_residual")
    if check_diag == "defer":
        return out, dmin
    if check_diag:
        assert dmin.item() >= 0  # same contract as the reference's assert (it also syncs)
    return out


# ------------------------------------------------------------------------------------------- rules 10/11
def mm_attention_rules(R_ss, R_qq, cam_sq, R_qs=None, apply_normalization=True, apply_self_in_rule_10=True,
                       nan_to_zero=False, check_diag=True):
    """Returns ``R_sq_addition`` (and ``R_ss_addition`` when ``R_qs`` is given)."""
    _dev(R_ss, R_qq, cam_sq, R_qs)
    R_ss, R_qq, cam_sq = _f32c(R_ss), _f32c(R_qq), _f32c(cam_sq)
    ns, nq = cam_sq.shape[-2], cam_sq.shape[-1]
    if R_ss.shape[-1] != ns or R_qq.shape[-1] != nq:
        raise MMXError("mm_attention_rules: R_ss %s, R_qq %s, cam_sq %s" % (tuple(R_ss.shape), tuple(R_qq.shape), tuple(cam_sq.shape)))
    flags = (_lib.MM_NORMALIZE if apply_normalization else 0) | (_lib.MM_SELF_IN_RULE10 if apply_self_in_rule_10 else 0) \
        | (_lib.MM_NAN_TO_ZERO if nan_to_zero else 0)
    sq_add = torch.empty(ns, nq, dtype=torch.float32, device=cam_sq.device)
    ss_add = None
    if R_qs is not None:
        R_qs = _f32c(R_qs)
        ss_add = torch.empty(ns, ns, dtype=torch.float32, device=cam_sq.device)
    want_diag = check_diag and apply_normalization      # the reference asserts in handle_residual even when the
    dmin = torch.empty(1, dtype=torch.float32, device=cam_sq.device) if want_diag else None   # self terms are unused
    need = lib().mmx_mm_rules_workspace_bytes(ns, nq)
    ws = _workspace(need, cam_sq.device)
    check(lib().mmx_mm_attention_rules(_p(R_ss), _p(R_qq), _p(R_qs), _p(cam_sq), _p(sq_add), _p(ss_add), ns, nq,
                                       flags, _p(dmin), _p(ws), need, _stream()), "mmx_mm_attention_rules")
    if want_diag:
        assert dmin.item() >= 0
    return (sq_add, ss_add) if R_qs is not None else sq_add


LXMERT_FUSED_MAX_TOKENS = 48


def lxmert_schedule(lang, vis, x_lang_cross, x_img_cross, x_lang_self, x_img_self, apply_normalization=True,
                    apply_self_in_rule_10=True, check_diag=True, text_len=None):
    """The whole LXMERT rule schedule in one launch.  Every argument is a list of ``(attn, grad)`` pairs of fp32
    ``[B, H, Nq, Nk]`` tensors (``x_img_*`` may omit the last cross layer).  Returns ``(R_tt, R_ti, R_ii, R_it)``
    with a leading batch dim.

    ``text_len`` (``[B]`` int tensor, optional): the batch is padded to ``T`` question tokens and sample ``b`` has
    ``text_len[b]`` real ones; its rules run on that leading block, the rest of its outputs is zero.
    ``check_diag``: ``True`` asserts the reference's ``handle_residual`` contract right away (one device->host read, like
    the reference's own assert); ``"defer"`` returns the device word as a fifth value instead -- nothing synchronises,
    the call can be captured into a hipGraph, and the caller asserts ``diag_min >= 0`` when it reads the results."""
    groups = [lang, vis, x_lang_cross, x_img_cross, x_lang_self, x_img_self]
    flat = [t for grp in groups for pair in grp for t in pair]
    _dev(*flat)
    if any(t.dtype != torch.float32 for t in flat):
        raise MMXError("lxmert_schedule: fp32 capture slabs only")
    keep = [[(a.contiguous(), g.contiguous()) for a, g in grp] for grp in groups]
    a0 = keep[2][0][0]                                     # [B, H, T, I]
    B, H, T, I = a0.shape
    n_x = len(keep[2])
    if len(keep[4]) != n_x or len(keep[3]) < n_x - 1 or len(keep[5]) < n_x - 1:
        raise MMXError("lxmert_schedule: inconsistent cross-layer lists")
    dev = a0.device
    R_tt = torch.empty(B, T, T, dtype=torch.float32, device=dev)
    R_ti = torch.empty(B, T, I, dtype=torch.float32, device=dev)
    R_ii = torch.empty(B, I, I, dtype=torch.float32, device=dev)
    R_it = torch.empty(B, I, T, dtype=torch.float32, device=dev)
    want_diag = bool(check_diag) and apply_normalization and apply_self_in_rule_10
    dmin = torch.empty(1, dtype=torch.float32, device=dev) if want_diag else None
    if text_len is not None:
        _dev(text_len)
        text_len = text_len.to(dtype=torch.int32).contiguous()
        if text_len.numel() != B:
            raise MMXError("lxmert_schedule: text_len needs one entry per sample")
    tables, alive = [], []
    for grp in keep:
        for which in (0, 1):
            tbl, arr = _lib.ptr_table([pair[which].data_ptr() for pair in grp])
            tables.append(tbl)
            alive.append(arr)
    flags = (_lib.MM_NORMALIZE if apply_normalization else 0) | (_lib.MM_SELF_IN_RULE10 if apply_self_in_rule_10 else 0)
    la, lg, va, vg, xlca, xlcg, xica, xicg, xlsa, xlsg, xisa, xisg = tables
    # two phases in one launch: chip-wide head averages, the last arriver of a sample runs the rules on the MFMA
    need = lib().mmx_lxmert_schedule_workspace_bytes(len(keep[0]), len(keep[1]), n_x, B, T, I)
    ws = _workspace(need, dev, tag="lxmert_schedule")
    check(lib().mmx_lxmert_schedule(la, lg, len(keep[0]), va, vg, len(keep[1]), xlca, xlcg, xica, xicg, xlsa, xlsg,
                                    xisa, xisg, n_x, B, H, T, I, flags, _p(text_len), _p(R_tt), _p(R_ti), _p(R_ii),
                                    _p(R_it), _p(dmin), _p(ws), need, _stream()), "mmx_lxmert_schedule")
    if check_diag == "defer":
        return R_tt, R_ti, R_ii, R_it, dmin
    if want_diag:
        assert dmin.item() >= 0   # the reference's handle_residual assert
    return R_tt, R_ti, R_ii, R_it


# ------------------------------------------------------------------------------------------- rollout
def rollout_chain(layers, normalize):
    """``layers``: list of ``[B, N, N]`` (or ``[N, N]``) fp32 maps, already sliced to ``start_layer:``."""
    _dev(*layers)
    mats = [_f32c(m) for m in layers]
    n = mats[0].shape[-1]
    batch = mats[0].numel() // (n * n)
    out = torch.empty(batch, n, n, dtype=torch.float32, device=mats[0].device)
    need = lib().mmx_rollout_workspace_bytes(batch, n)
    ws = _workspace(need, mats[0].device)
    tbl, _k = _lib.ptr_table([m.data_ptr() for m in mats])
    check(lib().mmx_rollout_chain(tbl, len(mats), batch, n, int(normalize), _p(out), _p(ws), need, _stream()),
          "mmx_rollout_chain")
    return out


# ------------------------------------------------------------------------------------------- attention capture
def _bhnd_strides(t, layout):
    """Element strides (batch, head, token) of a ``[B, N, H, D]``-viewable tensor with contiguous D."""
    if t.stride(-1) != 1:
        raise MMXError("attention operands need a contiguous head_dim")
    if layout == "bnhd":
        return t.stride(0), t.stride(2), t.stride(1)
    if layout == "bhnd":
        return t.stride(0), t.stride(1), t.stride(2)
    raise MMXError("layout %r" % layout)


def attn_capture_fwd(q, k, v, probs_out, scale, scale_mode=_lib.SCALE_Q_FIRST, mask=None, layout="bnhd", mma_bf16=False):
    """``q``: ``[B, Nq, H, D]`` view (``layout='bnhd'``) or ``[B, H, Nq, D]``; writes P into ``probs_out``
    (``[B, H, Nq, Nk]`` fp32 contiguous, caller-owned slab) and returns O in the same layout as q.
    ``mma_bf16``: products on the bf16 matrix cores (``MMX_ATTN_MMA_BF16``: operands rounded to bf16, fp32 accumulate
    and softmax; long-sequence streaming kernels only)."""
    _dev(q, k, v, probs_out, mask)
    if q.dtype != torch.float32 or k.dtype != torch.float32 or v.dtype != torch.float32:
        raise MMXError("attention capture op is fp32 in this ABI version")
    if layout == "bnhd":
        B, Nq, H, D = q.shape
        Nk = k.shape[1]
    else:
        B, H, Nq, D = q.shape
        Nk = k.shape[2]
    if probs_out.dtype not in _DTYPES or not probs_out.is_contiguous() or probs_out.numel() != B * H * Nq * Nk:
        raise MMXError("probs_out must be a contiguous fp32 / fp16 / bf16 [B,H,Nq,Nk] slab")
    o = torch.empty(q.shape, dtype=torch.float32, device=q.device)
    msb = msq = 0
    if mask is not None:
        mask = _f32c(mask)
        if mask.dim() == 2:          # [Nq, Nk]
            msb, msq = 0, Nk
        elif mask.dim() == 3:        # [B, Nq or 1, Nk]
            msb = mask.shape[1] * Nk if mask.shape[0] > 1 else 0
            msq = Nk if mask.shape[1] > 1 else 0
        else:
            raise MMXError("mask must be [Nq,Nk] or [B,Nq|1,Nk]")
    check(lib().mmx_attn_capture_fwd_ex(_p(q), _p(k), _p(v), *_bhnd_strides(q, layout), *_bhnd_strides(k, layout),
                                        *_bhnd_strides(v, layout), _p(mask), msb, msq, _p(probs_out),
                                        _DTYPES[probs_out.dtype] | (_lib.MMX_ATTN_MMA_BF16 if mma_bf16 else 0),
                                        _p(o), *_bhnd_strides(o, layout), B, H, Nq, Nk, D,
                                        float(scale), scale_mode, _stream()), "mmx_attn_capture_fwd")
    return o


def head_kernel_shape(n_q, n_k, head_dim, slab_dtype=torch.float32):
    """Does the register-resident whole-head backward (``attention_head.hip``) serve this shape?  (Its eligibility test, host side:
    what decides whether a bf16 gradient stream can pass through an exact-fp32 attention without conversion passes.)"""
    if not (slab_dtype == torch.float32 and 1 <= n_k <= 128 and 1 <= n_q <= 256 and head_dim % 4 == 0 and head_dim <= 64):
        return False
    # the kernel's own exclusions (attention_head.hip, attn_bwd_head_try): its LDS image must fit a CU, and the widest key
    # side is not instantiated beyond 8 query strips
    dp, ntk, ntq = (32 if head_dim <= 32 else 64), (n_k + 15) // 16, (n_q + 15) // 16
    lds = 4 * max(ntk * 16 * (2 * dp + 12), ntq * 16 * (ntk * 16 + 4 + dp + 4))
    return lds <= 160 * 1024 and not (ntk >= 7 and ntq > 8)


def attn_capture_bwd(q, k, v, probs, d_o, dprobs_out, scale, scale_mode=_lib.SCALE_Q_FIRST, need_dqkv=True,
                     layout="bnhd", out=None, batch=None, o=None, mma_bf16=False, rel_row=None):
    """Writes dP into ``dprobs_out`` and returns ``(dq, dk, dv)`` (``None`` when ``need_dqkv`` is False).
    ``out=(dq, dk, dv)`` lets the caller hand in (strided) views, e.g. of one packed dqkv tensor.
    ``batch``: shared-forward mode -- q/k/v/probs come from ONE forward (batch 1) and are broadcast (stride 0) over the
    ``batch`` upstream gradients in ``d_o``; dq/dk/dv/dprobs are per sample.
    ``o``: the forward's output (same layout as ``q``), optional: saves the long-sequence kernels a sweep over the keys.
    ``rel_row`` (``[B, N]`` fp32; row-relevancy mode, ``mmx_attn_capture_bwd_rowrel``): returns
    ``(dq, dk, dv, rel_row + rel_row . mean_h clamp(dP * P, 0))``; ``dprobs_out`` may then be ``None`` (dP is not stored)."""
    _dev(q, k, v, probs, d_o, dprobs_out, rel_row)
    if probs.dtype not in _DTYPES or (dprobs_out is not None and probs.dtype != dprobs_out.dtype):
        raise MMXError("attn_capture_bwd: probs / dprobs slabs must share one of fp32 / fp16 / bf16")
    if dprobs_out is None and rel_row is None:
        raise MMXError("attn_capture_bwd: no dprobs slab (only the row-relevancy mode runs without one)")
    if layout == "bnhd":
        B, Nq, H, D = q.shape
        Nk = k.shape[1]
    else:
        B, H, Nq, D = q.shape
        Nk = k.shape[2]
    shared = batch is not None and batch != B
    if shared:
        if B != 1:
            raise MMXError("shared-forward backward needs batch-1 q/k/v")
        B = batch
        q, k, v = (t.expand(B, *t.shape[1:]) for t in (q, k, v))      # stride-0 views, no copy
        if o is not None:
            o = o.expand(B, *o.shape[1:])
    probs_sb = 0 if shared else H * Nq * Nk
    if d_o.stride(-1) != 1:
        d_o = d_o.contiguous()
    io_bf16 = d_o.dtype == torch.bfloat16            # bf16 gradient stream: dq / dk / dv come back as bf16 too
    if io_bf16 and not mma_bf16 and not head_kernel_shape(Nq, Nk, D, probs.dtype):
        raise MMXError("attn_capture_bwd: a bf16 d_o needs mma_bf16=True (MMX_ATTN_IO_BF16) or a whole-head shape "
                       "(fp32 slabs, Nk <= 128, Nq <= 256, head_dim % 4 == 0 and <= 64: exact-fp32 arithmetic on the bf16 stream)")
    flags = (_lib.MMX_ATTN_MMA_BF16 if mma_bf16 else 0) | (_lib.MMX_ATTN_IO_BF16 if io_bf16 else 0)
    dq = dk = dv = None
    ws = None
    need = 0
    if need_dqkv:
        if out is not None:
            dq, dk, dv = out
        else:
            dq, dk, dv = (torch.empty(tuple(t.shape), dtype=d_o.dtype, device=t.device) for t in (q, k, v))
        need = lib().mmx_attn_capture_bwd_workspace_bytes(B, H, Nq)
        ws = _workspace(need, q.device, "attn_bwd")
    zero3 = (0, 0, 0)
    if o is not None and (o.dtype != torch.float32 or o.stride(-1) != 1):
        o = None                                               # only a hint: fall back to the two-sweep form
    if rel_row is not None:
        if not mma_bf16 or rel_row.dtype != torch.float32 or tuple(rel_row.shape) != (B, Nq) or Nq != Nk:
            raise MMXError("attn_capture_bwd: rel_row must be fp32 [B, N] of a self-attention, with mma_bf16=True")
        rel_row = rel_row.contiguous()
        rel_out = torch.empty_like(rel_row)
        need = lib().mmx_attn_capture_bwd_rowrel_workspace_bytes(B, H, Nq, Nk)
        ws = _workspace(need, q.device, "attn_bwd")
        check(lib().mmx_attn_capture_bwd_rowrel(
            _p(q), _p(k), _p(v), *_bhnd_strides(q, layout), *_bhnd_strides(k, layout), *_bhnd_strides(v, layout),
            _p(probs), probs_sb, _DTYPES[probs.dtype] | flags, _p(d_o), *_bhnd_strides(d_o, layout),
            _p(o), *(_bhnd_strides(o, layout) if o is not None else zero3), _p(dprobs_out), _p(dq), _p(dk), _p(dv),
            *(_bhnd_strides(dq, layout) if need_dqkv else zero3), *(_bhnd_strides(dk, layout) if need_dqkv else zero3),
            *(_bhnd_strides(dv, layout) if need_dqkv else zero3),
            B, H, Nq, Nk, D, float(scale), scale_mode, int(need_dqkv), _p(rel_row), _p(rel_out), _p(ws), need, _stream()),
            "mmx_attn_capture_bwd_rowrel")
        return dq, dk, dv, rel_out
    def call(d_o_, dq_, dk_, dv_, flags_):
        return lib().mmx_attn_capture_bwd_ex(
            _p(q), _p(k), _p(v), *_bhnd_strides(q, layout), *_bhnd_strides(k, layout), *_bhnd_strides(v, layout),
            _p(probs), probs_sb, _DTYPES[probs.dtype] | flags_, _p(d_o_), *_bhnd_strides(d_o_, layout),
            _p(o), *(_bhnd_strides(o, layout) if o is not None else zero3), _p(dprobs_out), _p(dq_), _p(dk_), _p(dv_),
            *(_bhnd_strides(dq_, layout) if need_dqkv else zero3), *(_bhnd_strides(dk_, layout) if need_dqkv else zero3),
            *(_bhnd_strides(dv_, layout) if need_dqkv else zero3),
            B, H, Nq, Nk, D, float(scale), scale_mode, int(need_dqkv), _p(ws), need, _stream())
    rc = call(d_o, dq, dk, dv, flags)
    if rc == _lib.MMX_ENOTSUP and io_bf16 and not mma_bf16:
        # the whole-head kernel is the only exact-fp32 kernel with bf16 gradient I/O; it turned this call down at run time (a view
        # it cannot load 8 bytes at a time, or option attn_head = 0): same arithmetic on an fp32 d_o, results rounded once to bf16
        f32 = [torch.empty(tuple(t.shape), dtype=torch.float32, device=t.device) for t in (q, k, v)] if need_dqkv else [None] * 3
        check(call(d_o.float(), *f32, flags & ~_lib.MMX_ATTN_IO_BF16), "mmx_attn_capture_bwd")
        if need_dqkv:
            for dst, src in zip((dq, dk, dv), f32):
                dst.copy_(src)
        return dq, dk, dv
    check(rc, "mmx_attn_capture_bwd")
    return dq, dk, dv


def attn_relprop(q, k, v, probs, o, cam_o, scale, scale_mode=_lib.SCALE_Q_FIRST, layout="bnhd", phase=3, cam_scores=None):
    """LRP relevance through the attention core (``mmx_attn_relprop[_phase]``; the two halved matmul / ``einsum`` relprops of
    DETR/modules/layers.py:770-781, lxmert_lrp.py:422-461, BERT_ours.py:345-395).  ``q [B, Nq, H, D]``, ``k`` / ``v [B, Nk,
    H, D]``, ``o`` / ``cam_o [B, Nq, H, D]`` (strided views ok), ``probs [B, H, Nq, Nk]`` fp32 contiguous.  Returns ``(cam_probs
    [B, H, Nq, Nk], cam_q, cam_k, cam_v)`` with the cams of q / k / v contiguous in ``layout``.

    ``phase``: ``_lib.LRP_VALUES`` -> ``(cam_probs, None, None, cam_v)`` from ``cam_o``; ``_lib.LRP_SCORES`` -> ``(None, cam_q,
    cam_k, None)`` from ``cam_scores [B, H, Nq, Nk]`` (the relevance of the pre-softmax scores; ``v / probs / o / cam_o`` may be
    ``None``); both (default) with ``cam_scores=None``: the fused core."""
    values, scores = bool(phase & _lib.LRP_VALUES), bool(phase & _lib.LRP_SCORES)
    given = [t for t in (q, k, v, probs, o, cam_o, cam_scores) if t is not None]
    _dev(*given)
    if any(t.dtype != torch.float32 for t in given):
        raise MMXError("attn_relprop: fp32 operands only")
    if layout == "bnhd":
        B, Nq, H, D = q.shape
        Nk = k.shape[1]
    else:
        B, H, Nq, D = q.shape
        Nk = k.shape[2]
    for name, t in (("probs", probs if values else None), ("cam_scores", cam_scores)):
        if t is not None and (not t.is_contiguous() or t.numel() != B * H * Nq * Nk):
            raise MMXError("attn_relprop: %s must be a contiguous [B,H,Nq,Nk] slab" % name)
    if scores and not values and cam_scores is None:
        raise MMXError("attn_relprop: the scores phase alone needs cam_scores")
    if values and cam_o.stride(-1) != 1:
        cam_o = cam_o.contiguous()
    new = lambda t: torch.empty(tuple(t.shape), dtype=torch.float32, device=q.device)          # noqa: E731
    cam_probs = torch.empty(B, H, Nq, Nk, dtype=torch.float32, device=q.device) if values else None
    cam_q, cam_k = (new(q), new(k)) if scores else (None, None)
    cam_v = new(v) if values else None
    zero3 = (0, 0, 0)
    st = lambda t: zero3 if t is None else _bhnd_strides(t, layout)           # noqa: E731
    check(lib().mmx_attn_relprop_phase(_p(q), _p(k), _p(v), _p(o), _p(cam_o), *st(q), *st(k), *st(v), *st(o), *st(cam_o),
                                       _p(probs), _p(cam_probs), _p(cam_q), _p(cam_k), _p(cam_v), *st(cam_q), *st(cam_k),
                                       *st(cam_v), B, H, Nq, Nk, D, float(scale), scale_mode, _p(cam_scores), int(phase),
                                       _stream()), "mmx_attn_relprop_phase")
    return cam_probs, cam_q, cam_k, cam_v


def detr_decoder_rows(self_pairs, cross_pairs, targets, shared_attn=False):
    """K2-DETR (``mmx_detr_decoder_rows``): the decoder half of DETR's rules for ROWS of ``R_q_i`` in four launches.
    ``self_pairs[l]`` / ``cross_pairs[l]``: ``(attn, grad)`` of decoder layer l's self- / cross-attention, fp32, gradient
    slabs ``[K*H, Q, *]``, probability slabs the same or ``[H, Q, *]`` when ``shared_attn`` (one forward for the K samples).
    ``targets``: ``[K]`` long.  Returns ``(s [K, Ni], diag_min [1])`` -- see include/mmx_relevancy.h."""
    targets = targets.reshape(-1).to(torch.long).contiguous()
    K = targets.numel()
    tens = [t for pair in list(self_pairs) + list(cross_pairs) for t in pair]
    _dev(targets, *tens)
    if any(t.dtype != torch.float32 or not t.is_contiguous() for t in tens):
        raise MMXError("detr_decoder_rows: fp32 contiguous slabs only")
    g0, c0 = self_pairs[0][1], cross_pairs[0][1]
    Q, Ni = g0.shape[-1], c0.shape[-1]
    H = g0.numel() // (K * Q * Q)
    if H * K * Q * Q != g0.numel() or c0.numel() != K * H * Q * Ni:
        raise MMXError("detr_decoder_rows: slab shapes %s / %s do not fit K = %d" % (tuple(g0.shape), tuple(c0.shape), K))
    L = len(self_pairs)
    s_out = torch.empty(K, Ni, dtype=torch.float32, device=g0.device)
    dmin = torch.empty(1, dtype=torch.float32, device=g0.device)
    need = lib().mmx_detr_decoder_rows_workspace_bytes(L, K, Q, Ni)
    ws = _workspace(need, g0.device, "detr_rows")
    tables = [_lib.ptr_table([pair[w].data_ptr() for pair in grp]) for grp in (self_pairs, cross_pairs) for w in (0, 1)]
    (sa, _k0), (sg, _k1), (ca, _k2), (cg, _k3) = tables
    check(lib().mmx_detr_decoder_rows(sa, sg, ca, cg, L, K, H, Q, Ni, 0 if shared_attn else H * Q * Q,
                                      0 if shared_attn else H * Q * Ni, _p(targets), _p(s_out), _p(dmin), _p(ws), need,
                                      _stream()), "mmx_detr_decoder_rows")
    return s_out, dmin
