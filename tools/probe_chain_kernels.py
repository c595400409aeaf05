"""Kernel-only probe (GPU box): the chain kernels against each other -- layer groups with barrier-free stream waves
(relevancy_chain_groups.hip, K1g: the fp32 default), the fused kernel of rounds 1-4 (K1), the strict-order / column-split kernel
(relevancy_chain_cols.hip, K1c) -- at the cfg-2 shapes over ROTATING slab sets (> 600 MB per tower: every byte from HBM), graph
replay, HIP events.  Variants: groups / workgroups per sample, nt policy, and the phase-skip debug flags.  (Round 5's relay kernel was
probed with this script too -- then tools/probe_chain_relay.py -- before it was removed: profiles/r05_chain_relay_probe.txt.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer_mm_explainability_amd import ops


def timed(fns, iters=10, warm=2):
    for _ in range(warm):
        for f in fns:
            f()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for f in fns:
            f()
    graph.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        graph.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (iters * len(fns)) * 1e3


DEFAULTS = {"self_chain_cols_c": 0, "self_chain_cols_nb": 0, "self_chain_algo": 0, "self_chain_groups": 0, "self_chain_nt": 1,
            "debug_flags": 0}


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    variants = [
        ("default (auto groups, groups kernel)", {}),
        ("per-sample G=1", {"self_chain_algo": 1, "self_chain_groups": 1}),
        ("fused kernel G=4 (r01-r04 default)", {"self_chain_algo": 1, "self_chain_groups": 4}),
        ("fused kernel G=3", {"self_chain_algo": 1, "self_chain_groups": 3}),
        ("fused kernel G=2", {"self_chain_algo": 1, "self_chain_groups": 2}),
        ("groups kernel G=4", {"self_chain_groups": 4}),
        ("groups kernel G=3", {"self_chain_groups": 3}),
        ("groups kernel G=2", {"self_chain_groups": 2}),
        ("groups kernel G=3, single-arriver combine (debug 16)", {"self_chain_groups": 3, "debug_flags": 16, "_cmp": "groups kernel G=3"}),
        ("groups kernel G=4, single-arriver combine (debug 16)", {"self_chain_groups": 4, "debug_flags": 16, "_cmp": "groups kernel G=4"}),
        ("groups kernel G=2, single-arriver combine (debug 16)", {"self_chain_groups": 2, "debug_flags": 16, "_cmp": "groups kernel G=2"}),
        ("groups kernel G=4 nt=0", {"self_chain_groups": 4, "self_chain_nt": 0}),
        ("groups kernel G=4 hand-off, no combine (debug 8)", {"self_chain_groups": 4, "debug_flags": 8}),
        ("groups kernel G=4 no combine (debug 1)", {"self_chain_groups": 4, "debug_flags": 1}),
        ("groups kernel G=3 no combine (debug 1)", {"self_chain_groups": 3, "debug_flags": 1}),
        ("groups kernel G=4 no MFMA no combine (debug 5)", {"self_chain_groups": 4, "debug_flags": 5}),
        ("fused kernel G=4 no combine (debug 1)", {"self_chain_algo": 1, "self_chain_groups": 4, "debug_flags": 1}),
        ("cols auto (C=1: the default for one group)", {"self_chain_algo": 5}),
        ("cols C=4", {"self_chain_algo": 5, "self_chain_cols_c": 4}),
        ("cols C=4 no rotation (debug 16)", {"self_chain_algo": 5, "self_chain_cols_c": 4, "debug_flags": 16}),
        ("cols C=2", {"self_chain_algo": 5, "self_chain_cols_c": 2}),
        ("cols C=3", {"self_chain_algo": 5, "self_chain_cols_c": 3}),
        ("cols C=1", {"self_chain_algo": 5, "self_chain_cols_c": 1}),
        ("cols NB=2", {"self_chain_algo": 5, "self_chain_cols_nb": 2}),
        ("cols NB=3", {"self_chain_algo": 5, "self_chain_cols_nb": 3}),
        ("cols C=4 stream only (debug 4)", {"self_chain_algo": 5, "self_chain_cols_c": 4, "debug_flags": 4}),
        ("cols C=4 stream only, no rotation (debug 20)", {"self_chain_algo": 5, "self_chain_cols_c": 4, "debug_flags": 20}),
    ]
    for (L, H, N, name, sets) in [(12, 8, 77, "txt", 3), (12, 12, 50, "img", 4)]:
        slabs = []
        for _ in range(sets):
            attn = [torch.rand(B * H, N, N, device="cuda").softmax(-1) for _ in range(L)]
            grad = [torch.randn(B * H, N, N, device="cuda") * 0.01 for _ in range(L)]
            slabs.append((attn, grad))
        nbytes = 2 * L * B * H * N * N * 4
        ref = None
        outs = {}
        for label, opts in variants:
            opts = dict(opts)
            cmp_to = opts.pop("_cmp", None)
            for k, v in {**DEFAULTS, **opts}.items():
                ops.set_option(k, v)
            try:
                rot = timed([(lambda a=a, g=g: ops.relevancy_self_chain(a, g, B)) for a, g in slabs])
                same = timed([lambda: ops.relevancy_self_chain(slabs[0][0], slabs[0][1], B)] * sets)
                out = ops.relevancy_self_chain(slabs[0][0], slabs[0][1], B).clone()
            except Exception as exc:                                   # a variant that does not launch must not end the probe
                print(f"{name}: {label:40s} FAILED: {exc}")
                continue
            outs[label] = out
            if label.startswith("per-sample"):
                ref = out
            same_bits = "" if ref is None or opts.get("debug_flags") else f" | == per-sample G=1: {bool(torch.equal(out, ref))}"
            if cmp_to:
                same_bits = f" | == {cmp_to}: {bool(torch.equal(out, outs[cmp_to]))}"
            print(f"{name} B={B}: {label:40s} rotating {rot:6.1f} us = {nbytes/rot/1e6:.3f} TB/s ({nbytes/rot/8e6:.3f} of 8 TB/s) | "
                  f"same buffers {same:6.1f} us{same_bits}", flush=True)
        del slabs
        torch.cuda.empty_cache()
    for k, v in DEFAULTS.items():
        ops.set_option(k, v)


if __name__ == "__main__":
    main()
