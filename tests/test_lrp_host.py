"""CPU suite: the closed-form LRP rules of ``lrp.py`` against outputs of the reference's own LRP layer library
(``DETR/modules/layers.py``, fixture ``lrp_layers.npz`` made by ``tests/golden/make_golden.py::gen_lrp_layers``).
The attention core runs on the plain-torch referee here (``oracle/lrp_torch.py``); the HIP kernels that replace it on the
product path are pinned on the same fixture in ``tests/test_gpu_lrp.py``."""
import numpy as np
import torch

from oracle import lrp_torch as lrp_oracle
from transformer_mm_explainability_amd import lrp


def close(a, b, atol=1e-6):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    np.testing.assert_allclose(a, b, rtol=1e-5, atol=atol)


def t(x):
    return torch.from_numpy(np.asarray(x))


def test_linear_add_clone_index_select_rules(golden):
    g = golden("lrp_layers")
    close(lrp.linear_relprop(t(g["lin_r"]), t(g["lin_x"]), t(g["lin_w"])), g["lin_out"])
    ra, rb = lrp.add_relprop(t(g["add_r"]), t(g["add_a"]), t(g["add_b"]))
    close(ra, g["add_out_a"])
    close(rb, g["add_out_b"])
    close(lrp.clone_relprop(list(t(g["clone_r"])), t(g["clone_x"])), g["clone_out"])
    close(lrp.index_select_relprop(t(g["sel_r"]), t(g["sel_x"]), 0, torch.tensor([2])), g["sel_out"])


def mha_tape(g, tag):
    """The forward of DETR/modules/layers.py:728-768 restated on the fixture's weights -> the tape ``mha_relprop`` reads."""
    H = int(g[tag + "_heads"])
    w = {k[len(tag) + 4:]: t(v) for k, v in g.items() if k.startswith(tag + "_w__")}
    bf = lambda x: t(x).permute(1, 0, 2)                                     # noqa: E731
    query, key, value = bf(g[tag + "_query"]), bf(g[tag + "_key"]), bf(g[tag + "_value"])
    B, T, E = query.shape
    D = E // H
    lin = lambda x, n: torch.nn.functional.linear(x, w[n + ".weight"], w[n + ".bias"])      # noqa: E731
    q, k, v = (lin(x, n).view(B, -1, H, D) for x, n in ((query, "q_proj"), (key, "k_proj"), (value, "v_proj")))
    scale = float(D) ** -0.5
    probs = torch.softmax(torch.einsum("bthd,bshd->bhts", q * scale, k), dim=-1)
    o = torch.einsum("bhts,bshd->bthd", probs, v)
    close(probs.reshape(B * H, T, -1), g[tag + "_attn"], atol=1e-6)
    tape = dict(query=query, key=key, value=value, q=q, k=k, v=v, o=o, probs=probs, scale=scale)
    return tape, (w["q_proj.weight"], w["k_proj.weight"], w["v_proj.weight"], w["out_proj.weight"])


def test_mha_relprop_matches_reference(golden):
    g = golden("lrp_layers")
    for tag in ("mha", "mha0"):                                             # mha0: zero value stream -> the rescale branch
        tape, weights = mha_tape(g, tag)
        cam_q, cam_k, cam_v, cam_p = lrp.mha_relprop(t(g[tag + "_cam_out"]).permute(1, 0, 2), tape, weights,
                                                     lrp_oracle.detr_core(tape))
        B, H, T, S = cam_p.shape
        close(cam_p.reshape(B * H, T, S), g[tag + "_attn_cam"], atol=2e-6)
        close(cam_q.permute(1, 0, 2), g[tag + "_cam_q"], atol=2e-6)
        close(cam_k.permute(1, 0, 2), g[tag + "_cam_k"], atol=2e-6)
        close(cam_v.permute(1, 0, 2), g[tag + "_cam_v"], atol=2e-6)
    assert float(np.abs(g["mha0_cam_v"]).max()) == 0.0 and float(np.abs(g["mha0_cam_q"]).max()) > 0.0
