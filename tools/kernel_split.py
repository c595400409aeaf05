"""Kernel-time split of a rocprofv3 kernel trace (rocpd sqlite) by kernel family -> JSON for ``bench.py``'s ``roofline_step``.

    python tools/kernel_split.py <results.db> <steps in the trace> > profiles/rNN_bench_kernel_split.json

Families: library GEMMs (rocBLAS / hipBLASLt ``Cijk_*``), our attention-capture kernels, our chain kernels, our fused
elementwise kernels (QuickGELU, LayerNorm forward / backward), ATen LayerNorm, ATen elementwise / copies / reductions.
The trace should be taken with ``bench.py --headline-only`` so that it holds headline steps (plus the two graph warm-ups).
"""
import json
import sqlite3
import sys

db, steps = sys.argv[1], int(sys.argv[2])
rows = sqlite3.connect(db).execute("select name, count(*), sum(end-start) from kernels group by name").fetchall()


def family(name):
    if name.startswith("Cijk_") or "rocblas" in name.lower() or "hipblaslt" in name.lower():
        return "library_gemm_fp32"
    if "mmx::attn" in name:
        return "mmx_attention_capture"
    if "mmx::self_chain" in name or "mmx::avg_heads" in name or "mmx::bmm" in name or "mmx::row_normalise" in name:
        return "mmx_relevancy_chain"
    if "mmx::" in name:
        return "mmx_fused_elementwise"
    if "layer_norm" in name:
        return "aten_layernorm"
    return "aten_elementwise_other"


agg, calls = {}, {}
for name, n, ns in rows:
    f = family(name)
    agg[f] = agg.get(f, 0) + ns
    calls[f] = calls.get(f, 0) + n
total = sum(agg.values())
out = {"trace_steps": steps, "kernel_ms_per_step": round(total / 1e6 / steps, 3),
       "families": {f: {"percent": round(100.0 * v / total, 2), "ms_per_step": round(v / 1e6 / steps, 3),
                        "launches_per_step": round(calls[f] / steps, 1)} for f, v in sorted(agg.items(), key=lambda kv: -kv[1])}}
print(json.dumps(out, indent=1))
