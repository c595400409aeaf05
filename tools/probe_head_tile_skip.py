"""Whole-head attention kernels (attention_head.hip) at the cfg-2 TEXT shape (causal mask) with and without the masked-tile skip
(option attn_head_tile_skip), interleaved rounds in one process: us per launch over rotating buffer sets (tools/probe_head_attention.py)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import probe_head_attention as ph  # noqa: E402

ops = ph.ops
res = {}
for rnd in range(3):
    for skip in (0, 1):
        ops.set_option("attn_head_tile_skip", skip)
        for leg, (t, nbytes) in ph.run_text().items():
            res.setdefault((leg, skip), []).append((t, nbytes))
ops.set_option("attn_head_tile_skip", 1)
for (leg, skip), v in sorted(res.items()):
    t = statistics.median(x[0] for x in v)
    print("text B=64 H=8 N=77 d=64  %-14s tile_skip=%d  median %6.2f us (min %.2f max %.2f)  %.2f TB/s = %4.1f %% of 8 TB/s"
          % (leg, skip, t, min(x[0] for x in v), max(x[0] for x in v), v[0][1] / t / 1e6, v[0][1] / t / 8e4))
