"""Probe: CPU oracle timing vs thread count on the GPU box's host (bounded by the caller's `timeout`)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
threads, B = int(sys.argv[1]), int(sys.argv[2])
torch.set_num_threads(threads)
from oracle import clip_torch
from transformer_mm_explainability_amd import clip_model
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
model = clip_model.random_init("ViT-B/32", 0)
sd = clip_torch.prepare_state_dict(model.state_dict(), 8)
image, texts = bench.synthetic_inputs(B, "cpu", 0)
for rep in range(2):
    tm = {}
    t0 = time.perf_counter()
    clip_torch.interpret(sd, image, texts, 0, 0, timings=tm)
    dt = time.perf_counter() - t0
    print(f"threads={threads} B={B} rep={rep}: {dt:.2f}s  {B/dt:.2f} maps/s  {tm}", flush=True)
