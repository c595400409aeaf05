"""Interleaved A/B of the text tower's chain launch with and without MMX_CHAIN_CAUSAL (rotating slab sets > Infinity Cache)."""
import statistics
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from transformer_mm_explainability_amd import ops

dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
for bb in (64, 128, 256):
    sets = 3 if bb == 64 else 2
    keep = [bench.synthetic_text_slabs(bb, dev, seed=10 + k) for k in range(sets)]
    res = {}
    plans = {c: [ops.ChainPlan(a, g, bb, causal=c) for a, g in keep] for c in (False, True)}
    assert torch.equal(plans[False][0].launch().clone(), plans[True][0].launch().clone())
    for rnd in range(5):
        for c in (False, True):
            st = {"i": 0}
            def fn():
                plans[c][st["i"] % sets].launch(); st["i"] += 1
            res.setdefault(c, []).append(bench.kernel_time_us(fn, 4 * sets + 1, stream))
    by = 2 * 12 * bb * 8 * 77 * 77 * 4 + bb * 77 * 77 * 4
    for c in (False, True):
        m = statistics.median(res[c])
        print("B=%d causal=%d  median %.2f us  min %.2f  max %.2f  alg GB/s %.0f" % (bb, c, m, min(res[c]), max(res[c]), by / m / 1e3))
    if bb <= 128:      # explicit group counts under the causal form (0 = the auto rule)
        res = {}
        for rnd in range(3):
            for G in ((0, 2, 3, 4) if bb == 64 else (0, 2)):
                ops.set_option("self_chain_groups", G)
                pl = [ops.ChainPlan(a, g, bb, causal=True) for a, g in keep]
                st = {"i": 0}
                def fn():
                    pl[st["i"] % sets].launch(); st["i"] += 1
                res.setdefault(G, []).append(bench.kernel_time_us(fn, 4 * sets + 1, stream))
        ops.set_option("self_chain_groups", 0)
        for G, v in res.items():
            print("B=%d causal G=%d  median %.2f us" % (bb, G, statistics.median(v)))
    del keep, plans
