"""A / B on one box: the cfg-5 step with the text tower's whole-head backward on the bf16 gradient stream (attention_head.hip, IOH)
vs the two conversion passes around it (round 3 / early round 4).  `python tools/probe_cfg5_head16.py [batch]`"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tools import bench_legs  # noqa: E402
from transformer_mm_explainability_amd import clip_explainability as ce, ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
model, image, texts, _, _ = bench_legs.cfg5_setup(B, torch.device("cuda"))
f = lambda: ce.interpret(image, texts, model, "cuda", start_layer=0, start_layer_text=0)  # noqa: E731
real = ops.head_kernel_shape


def step_ms(reps=6):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for rnd in range(2):
    for name, fn in (("bf16 I/O head kernel", real), ("conversion passes    ", lambda *a, **k: False)):
        ops.head_kernel_shape = fn
        ms = step_ms()
        print("round %d  %s : %.2f ms per step = %.1f maps/s" % (rnd, name, ms, B / ms * 1e3), flush=True)
ops.head_kernel_shape = real
