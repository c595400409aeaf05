#!/usr/bin/env python3
"""bench.py -- relevancy maps/s (forward + ONE backward + fused chain), CLIP ViT-B/32, batch 64 fp32.

Contract (see the task statement): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 the driver
launches one rank per GPU through ``torch.distributed.run``.  One "step" = one pass of the hot path over one
batch of 64 synthetic image<->text pairs per rank: CLIP forward with the HIP attention-capture op, one backward
that fills every layer's gradient slab, the fused relevancy chain for both towers (all 12+12 layers,
``start_layer=0``).  Rank 0 prints ONE JSON line.  Inputs are resident in HBM before the timed region.

Extra objects in the line:
  roofline      -- the chain kernel (text-tower instantiation, the longer one): algorithmic bytes / HIP-event time
                   measured on the stream the kernel runs on, against the 8 TB/s HBM peak
  roofline_step -- the WHOLE step against what really bounds it: the exact-fp32 matrix FLOPs the step executes (body
                   GEMMs + attention products + chain) / ms_per_step vs the 157.3 TFLOP/s fp32 MFMA peak, with the
                   kernel-time split of the committed rocprofv3 summary next to it
  cpu_baseline  -- the reference algorithm (oracle/clip_torch.py: per-layer autograd.grad like the notebook) on this
                  box's host cores, rank 0 / N=1 only, on a bounded sample
"""
from __future__ import annotations

import argparse
import ctypes as C
import glob
import json
import os
import shutil
import sys
import tempfile
import time

# multi-process GPU work on this driver stack needs dmabuf IPC (the task environment exports it; a bare shell may not)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def enable_tuned_gemms():
    """PyTorch-ROCm TunableOp with a committed, pre-tuned selection of hipBLASLt / rocBLAS solutions for the fp32 GEMM
    shapes of this workload on gfx950 (tuning itself is OFF here; shapes not in the file use the default heuristic;
    the file is ignored by PyTorch if its ROCm / hipBLASLt validators do not match the box).  +7 % maps/s measured.
    Must run before the first GEMM; one copy per device ordinal because PyTorch appends the ordinal to the file name."""
    src = os.path.join(ROOT, "transformer-mm-explainability_amd", "tuning", "tunableop_gfx950_clip_vitb32_b64.csv")
    if os.environ.get("PYTORCH_TUNABLEOP_ENABLED") is not None or not os.path.exists(src):
        return
    tmp = tempfile.mkdtemp(prefix="mmx_tunableop_")
    for ordinal in range(8):
        shutil.copy(src, os.path.join(tmp, "gemm%d.csv" % ordinal))
    os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
    os.environ["PYTORCH_TUNABLEOP_TUNING"] = "0"
    os.environ["PYTORCH_TUNABLEOP_FILENAME"] = os.path.join(tmp, "gemm.csv")


enable_tuned_gemms()
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy ceiling)
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, exact fp32 (no TF32 / xf32 on gfx950)
BATCH = 64
MODEL = "ViT-B/32"


def step_flops(batch):
    """Matrix FLOPs ONE headline step executes (CLIP ViT-B/32, all layers, shared image forward), by operation.  Row-wise
    products of the two top blocks run on one row per sample (``backward_tape(dy_rows=...)``); the lowest block has no
    in-projection backward.  Everything is exact fp32 on the MFMA (library GEMMs or our kernels)."""
    def tower(L, E, N, H, m_fwd, m_bwd):
        gemm_fwd = L * 2 * m_fwd * 12 * E * E                      # in_proj 3E^2 + out_proj E^2 + c_fc 4E^2 + c_proj 4E^2
        full, top, low = 2 * m_bwd * 12 * E * E, 2 * m_bwd * 3 * E * E + 2 * batch * 9 * E * E, 2 * m_bwd * 9 * E * E
        gemm_bwd = (L - 2) * full + top + low
        d = E // H
        attn_fwd = L * 4 * (m_fwd // N) * H * N * N * d
        attn_bwd = (L - 1) * 8 * (m_bwd // N) * H * N * N * d + 2 * (m_bwd // N) * H * N * N * d   # lowest block: dP only
        chain = L * ((m_bwd // N) * (2 * H * N * N + 2 * N * N * N))
        return {"gemm_fwd": gemm_fwd, "gemm_bwd": gemm_bwd, "attention": attn_fwd + attn_bwd, "chain": chain}
    img = tower(12, 768, 50, 12, 50, batch * 50)                   # forward once (shared), backward at batch B
    txt = tower(12, 512, 77, 8, batch * 77, batch * 77)
    out = {k: img[k] + txt[k] for k in img}
    out["total"] = sum(out.values())
    return out


BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 matrix-core peak (AMD's 5 PF figure is 2:1 sparse)
CFG5_BATCH = 128
CFG5_MODEL = "ViT-L/14@336"


def main_cfg5(args):
    """Optional leg (``--workload cfg5``; NOT the driver's default): BASELINE.json config 5's shape on this GPU -- CLIP
    ViT-L/14@336 with the bf16 body of ``clip_model.CLIP.set_body_dtype``, batch 128 per GPU, all layers, eager."""
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("LOCAL_RANK", 0), ("WORLD_SIZE", 1)))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    from tools import bench_legs
    from transformer_mm_explainability_amd import clip_explainability as ce
    model, image, texts, attn_layer, attn_flops = bench_legs.cfg5_setup(CFG5_BATCH, device, rank)
    attn_flops = attn_flops["algorithmic"]          # 4 products per attention backward (the kernel pair executes 5)
    n_img = 576
    row = n_img + 77 * 77
    gathered = torch.empty(world * CFG5_BATCH, row, device=device) if world > 1 else None
    packed = torch.empty(CFG5_BATCH, row, device=device) if world > 1 else None

    def step():
        R_text, R_image = ce.interpret(image, texts, model, device, start_layer=0, start_layer_text=0)
        if world > 1:
            packed[:, :n_img] = R_image
            packed[:, n_img:] = R_text.reshape(CFG5_BATCH, -1)
            dist.all_gather_into_tensor(gathered, packed)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms = elapsed / args.steps * 1e3
    fl = bench_legs.cfg5_step_flops(CFG5_BATCH)
    # our dominant kernel pair of this step: the bf16 attention backward of one image-tower layer (stand-alone launches)
    us = kernel_time_us(attn_layer, 5, torch.cuda.current_stream())
    if rank == 0:
        print(json.dumps({
            "metric": "relevancy maps/sec (fwd+bwd+rollout), CLIP ViT-L/14@336", "value": round(world * CFG5_BATCH / (ms * 1e-3), 2),
            "unit": "maps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE config 5 shape: CLIP ViT-L/14@336 (577 image tokens) image<->text relevancy, batch=128 "
                                   "per GPU, all 24+12 layers; bf16 body (GEMMs + image attention on the bf16 matrix cores, fp32 "
                                   "accumulation / LayerNorm / softmax / relevancy); random-init weights, synthetic inputs; eager",
                       "global_batch": world * CFG5_BATCH, "parallelism": "dp%d (independent batches, all-gather of maps)" % world,
                       "resident_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)},
            "roofline": {"bound": "mfma", "kernel": "attn_bwd_q_v3_kernel + attn_bwd_kv_v4_kernel (one image-tower layer, "
                                                    "row-relevancy mode)", "achieved": round(attn_flops / us / 1e6, 1),
                         "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(attn_flops / us / 1e6 / BF16_MFMA_PEAK_TFLOPS, 4),
                         "traffic": None, "us_per_launch": round(us, 1),
                         "note": "see profiles/ (newest rNN_cfg5_probe.txt) for the SQ counters of this pair"},
            "roofline_step": {"bound": "mfma", "achieved": round(fl["total"] / (ms * 1e-3) / 1e12, 1), "peak": BF16_MFMA_PEAK_TFLOPS,
                              "unit": "TFLOP/s", "frac": round(fl["total"] / (ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4),
                              "flop_per_step": fl},
            "cpu_baseline": None}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def synthetic_inputs(batch, device, seed):
    g = torch.Generator().manual_seed(1 + seed)
    image = torch.randn(1, 3, 224, 224, generator=g)
    texts = torch.zeros(batch, 77, dtype=torch.long)
    g2 = torch.Generator().manual_seed(2 + seed)
    for b in range(batch):
        n = int(torch.randint(3, 11, (1,), generator=g2))
        texts[b, 0] = 49406
        texts[b, 1:1 + n] = torch.randint(1, 49405, (n,), generator=g2)
        texts[b, 1 + n] = 49407                      # EOT must be the arg-max id (model.py:360)
    return image.to(device), texts.to(device)


def kernel_time_us(fn, iters, stream):
    """Average duration of the launches issued by ``fn`` with HIP events recorded on ``stream`` itself."""
    from transformer_mm_explainability_amd import _lib
    lib = _lib.lib()
    ev = [C.c_void_p(), C.c_void_p()]
    for e in ev:
        _lib.check(lib.mmx_event_create(C.byref(e)), "event_create")
    sp = C.c_void_p(stream.cuda_stream)
    fn()
    stream.synchronize()
    _lib.check(lib.mmx_event_record(ev[0], sp), "event_record")
    for _ in range(iters):
        fn()
    _lib.check(lib.mmx_event_record(ev[1], sp), "event_record")
    ms = C.c_float()
    _lib.check(lib.mmx_event_elapsed_ms(ev[0], ev[1], C.byref(ms)), "event_elapsed")
    for e in ev:
        lib.mmx_event_destroy(e)
    return ms.value * 1e3 / iters


CPU_THREADS_CAP = 32   # torch CPU autograd on >64 threads gets slower, not faster (256-thread box: 128 threads = 0.9 maps/s)
CPU_THREAD_CANDIDATES = (16, 32)   # the best of these is reported: which one wins depends on the box (profiles/r02_cpu_threads.txt: 32;
                                   # profiles/r05_cpu_threads.txt: 16 threads 9.2 maps/s vs 32 threads 3.5 at batch 8)


def cpu_baseline_worker(sample_b, reps):
    """Runs in a child process (bounded by a timeout in the parent): the reference algorithm on host cores."""
    from oracle import clip_torch
    from transformer_mm_explainability_amd import clip_model
    model = clip_model.random_init(MODEL, seed=0)       # parameters only; the CPU path never calls the HIP op
    sd = clip_torch.prepare_state_dict(model.state_dict(), 8)
    image, texts = synthetic_inputs(sample_b, "cpu", 0)
    best, tried = None, {}
    for cores in sorted({min(os.cpu_count() or 1, c) for c in CPU_THREAD_CANDIDATES}):
        torch.set_num_threads(cores)
        clip_torch.interpret(sd, image, texts, 0, 0)  # warm-up
        times = []
        for _ in range(reps):
            split = {}
            t0 = time.perf_counter()
            clip_torch.interpret(sd, image, texts, 0, 0, timings=split)
            times.append((time.perf_counter() - t0, split))
        times.sort(key=lambda t: t[0])
        med, split = times[len(times) // 2]
        tried[cores] = round(sample_b / med, 3)
        if best is None or med < best[0]:
            best = (med, split, cores)
    med, split, cores = best
    print(json.dumps({"value": round(sample_b / med, 3), "unit": "maps/s", "cores": cores, "kind": "port",
                      # forward / the 24 per-layer partial backwards / the rule chain of the median run
                      "split_s": {"forward": round(split["forward_s"], 3), "backward": round(split["backward_s"], 3),
                                  "rule_chain": round(split["rules_s"], 3), "total": round(med, 3)},
                      "maps_per_s_by_threads": tried,
                      "calibration": "profiles/r05_cpu_port_calibration.txt (this port vs the reference's own CLIP/clip/model.py + "
                                     "notebook cell 6, same weights, inputs and threads, run in the build container)",
                      "sample": "reference algorithm (hooked CLIP ViT-B/32 fwd + one autograd.grad per layer + rule "
                                "chain, all 12+12 layers) restated in oracle/clip_torch.py, torch fp32 CPU, the best of %s threads "
                                "(%d) of %d host cores, batch %d of the same synthetic workload, median of %d"
                                % ("/".join(str(c) for c in sorted(tried)), cores, os.cpu_count() or 1, sample_b, reps)}), flush=True)


def cpu_baseline(sample_b=16, reps=3, timeout_s=240):
    import subprocess
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(sample_b), str(reps)],
                             capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
        for ln in reversed(out.stdout.strip().splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
        return {"value": None, "unit": "maps/s", "cores": 0, "kind": "port", "sample": "worker failed: " + out.stderr[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "maps/s", "cores": 0, "kind": "port", "sample": "timed out after %ds" % timeout_s}


def cpu_leg_worker(name):
    """Child process: the reference algorithm of one BASELINE configuration on the host cores, on a bounded sample
    (the oracle bodies of oracle/: plain torch autograd, pinned on the reference's own code in tests/test_oracle_golden.py)."""
    cores = min(os.cpu_count() or 1, CPU_THREADS_CAP)
    torch.set_num_threads(cores)
    total = os.cpu_count() or 1

    def median(fn, reps):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2]

    if name == "cfg1":
        from oracle import vit_torch
        from transformer_mm_explainability_amd import vit_model
        torch.manual_seed(0)
        sd = dict(vit_model.vit_base_patch16_224().float().state_dict())
        x = torch.randn(1, 3, 224, 224)
        vit_torch.generate_relevance(sd, x, 12, 5)
        sec = median(lambda: vit_torch.generate_relevance(sd, x, 12, 5), 5)
        out = {"value": round(1 / sec, 3), "unit": "maps/s", "sample": "oracle/vit_torch.generate_relevance (ViT-B/16 forward + one "
               "backward + 12-layer rule chain, notebook cell 7), one 224x224 image, median of 5"}
    elif name == "cfg3":
        from oracle import detr_torch
        from transformer_mm_explainability_amd import detr_model
        torch.manual_seed(0)
        sd = detr_torch.prepare_state_dict(detr_model.detr_resnet50_head().state_dict())
        feats = torch.randn(1, 2048, 25, 38) * 0.5
        pos = detr_torch.position_embedding_sine(torch.zeros(1, 25, 38, dtype=torch.bool), 128, normalize=True)
        detr_torch.generate_ours(sd, feats, pos, [3], 8)
        sec = median(lambda: detr_torch.generate_ours(sd, feats, pos, [3], 8), 3)
        out = {"value": round(1 / sec, 3), "unit": "queries/s", "sample": "oracle/detr_torch.generate_ours (DETR-R50 transformer + "
               "heads forward at 950 image tokens + one backward + the matrix-route rules 6 / 7 / 10 incl. the 950^3 encoder "
               "chain), ONE kept query per call as DETR/mask_generator.py:90-110 runs it, median of 3"}
    elif name == "cfg4":
        from oracle import lxmert_torch
        from transformer_mm_explainability_amd import lxmert_model as lm
        torch.manual_seed(0)
        sd = lxmert_torch.prepare_state_dict(lm.LxmertForQuestionAnswering(lm.LxmertConfig()).state_dict())
        gb = torch.Generator().manual_seed(2)
        T, I, n = 14, 36, 4
        items = [dict(input_ids=torch.randint(1, 30000, (1, T), generator=gb), attention_mask=torch.ones(1, T),
                      token_type_ids=torch.zeros(1, T, dtype=torch.long), visual_feats=torch.randn(1, I, 2048, generator=gb),
                      visual_pos=torch.rand(1, I, 4, generator=gb)) for _ in range(n)]

        def sample(it):                      # explain + the 9 re-runs of the perturbation test (perturbation.py:85-194)
            lxmert_torch.generate_ours(sd, 12, it)
            with torch.no_grad():
                for _ in range(9):
                    lxmert_torch.forward(sd, 12, **it)
        sample(items[0])
        sec = median(lambda: [sample(it) for it in items], 3) / n
        out = {"value": round(1 / sec, 3), "unit": "samples/s", "sample": "oracle/lxmert_torch: generate_ours (LXMERT-base forward + "
               "backward + 38-rule schedule) + 9 further forwards per sample (the perturbation steps; region removal itself not "
               "restated), T = 14, I = 36, one item per call as perturbation.py runs it, %d items, median of 3" % n}
    elif name == "cfg5":
        from oracle import clip_torch
        from transformer_mm_explainability_amd import clip_model
        sd = clip_torch.prepare_state_dict(clip_model.random_init(CFG5_MODEL, seed=0).state_dict(), 12)
        g = torch.Generator().manual_seed(1)
        image = torch.randn(1, 3, 336, 336, generator=g)
        texts = torch.zeros(1, 77, dtype=torch.long)
        texts[0, 0], texts[0, 1:6], texts[0, 6] = 49406, torch.arange(1, 6), 49407
        t0 = time.perf_counter()
        clip_torch.interpret(sd, image, texts, 0, 0)
        sec = time.perf_counter() - t0
        out = {"value": round(1 / sec, 4), "unit": "maps/s", "sample": "oracle/clip_torch.interpret (the notebook's per-layer "
               "autograd.grad loop) on CLIP ViT-L/14@336 in fp32, ONE pair, all 24 + 12 layers, a single cold run"}
    else:
        raise SystemExit("unknown leg " + name)
    out.update(cores=cores, kind="port")
    out["sample"] += "; torch fp32 CPU, %d threads of %d host cores" % (cores, total)
    print(json.dumps(out), flush=True)


def cpu_leg(name, timeout_s=120):
    import subprocess
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-leg-worker", name], capture_output=True, text=True,
                             timeout=timeout_s, cwd=ROOT)
        for ln in reversed(out.stdout.strip().splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
        return {"value": None, "cores": 0, "kind": "port", "sample": "worker failed: " + out.stderr[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "cores": 0, "kind": "port", "sample": "timed out after %ds" % timeout_s}


def synthetic_text_slabs(batch, device, seed=3, layers=12, heads=8, n=77):
    """Synthetic slabs of the text tower's chain launch: causal softmax probabilities, random gradients."""
    g = torch.Generator(device=device).manual_seed(seed)
    mask = torch.full((n, n), float("-inf"), device=device).triu_(1)
    attn = [(torch.randn(batch * heads, n, n, device=device, generator=g) + mask).softmax(-1) for _ in range(layers)]
    grad = [torch.randn(batch * heads, n, n, device=device, generator=g) * 1e-2 for _ in range(layers)]
    return attn, grad


def causal_requested_bytes(layers, batch, heads, n):
    """Bytes the chain launch asks for under MMX_CHAIN_CAUSAL: the 16-byte chunks of the row-major n x n slabs that are NOT entirely
    above the diagonal (csrc/chain_stream.h: first element above it and the chunk does not wrap into the next row), both slabs, every
    head and layer, plus the R write.  (The memory system still moves whole 128-byte lines: `traffic` is the measured figure.)"""
    live = 0
    for c in range((n * n + 3) // 4):
        row, col = divmod(4 * c, n)
        live += not (col > row and col + 3 < n)
    return 2 * layers * batch * heads * live * 16 + batch * n * n * 4


def chain_worker(batch, launches):
    """Child process run UNDER rocprofv3 (``measure_chain_counters``): nothing but ``launches`` stand-alone launches of the text
    tower's chain kernel at ``batch`` (the kernel ``roofline`` is quoted on), so that the per-dispatch PMC values are that kernel's."""
    from transformer_mm_explainability_amd import ops
    dev = torch.device("cuda", 0)
    attn, grad = synthetic_text_slabs(batch, dev)
    plan = ops.ChainPlan(attn, grad, batch, causal=True)      # the text tower's launch as `interpret` makes it (causal mask)
    for _ in range(launches):
        plan.launch()
    torch.cuda.synchronize()


def _rocprofv3():
    import shutil
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    return exe if os.path.exists(exe) else None


def _csv_rows(directory, suffix):
    import csv
    for root, _, files in os.walk(directory):
        for name in files:
            if name.endswith(suffix):
                with open(os.path.join(root, name)) as f:
                    return list(csv.DictReader(f))
    return []


def measure_chain_counters(timeout_s=150):
    """HBM traffic of ONE text-tower chain launch, measured in THIS run: two child processes of this file under
    ``rocprofv3 --pmc FETCH_SIZE`` / ``--pmc WRITE_SIZE`` (separate passes, with --kernel-trace only, as MI355X_MICROARCH.md's HBM
    section prescribes; FETCH_SIZE doubled: gfx950 tallies the 128-B requests of wide coalesced reads at 64 B; the counters are in
    KB).  Returns a dict or None (no rocprofv3 on this box / a pass failed: the caller falls back to the committed file and says so)."""
    import subprocess
    import tempfile
    exe = _rocprofv3()
    if exe is None:
        return None
    out = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        with tempfile.TemporaryDirectory(prefix="mmx_pmc_", dir="/tmp") as tmp:
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "pmc", "--",
                   sys.executable, os.path.abspath(__file__), "--chain-worker", str(BATCH), "4"]
            try:
                run = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd="/tmp",
                                     env=dict(os.environ, TMPDIR="/tmp", PYTHONPATH=ROOT))
            except subprocess.TimeoutExpired:
                return None
            vals = [float(r["Counter_Value"]) for r in _csv_rows(tmp, "counter_collection.csv")
                    if "self_chain" in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter]
            if run.returncode != 0 or not vals:
                log("rocprofv3 --pmc %s pass failed (rc %d): %s" % (counter, run.returncode, run.stderr[-300:]))
                return None
            out[counter] = sorted(vals)[len(vals) // 2] * 1024.0          # median dispatch, KB -> bytes
    return {"fetch_bytes": int(out["FETCH_SIZE"] * 2), "write_bytes": int(out["WRITE_SIZE"]),
            "how": "measured in this run: two child processes of bench.py (--chain-worker: stand-alone launches of the same kernel on "
                   "synthetic slabs of the same shape) under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, "
                   "--kernel-trace only), median dispatch, FETCH_SIZE x2 (gfx950 correction), KB = 1024 B"}


def measure_chain_in_step(timeout_s=240):
    """The chain kernel INSIDE the replayed headline step, measured in this run: a child ``bench.py --headline-only`` under
    ``rocprofv3 --kernel-trace`` (CSV), average duration of the text tower's launch over the traced steps."""
    import subprocess
    import tempfile
    exe = _rocprofv3()
    if exe is None:
        return None
    with tempfile.TemporaryDirectory(prefix="mmx_trace_", dir="/tmp") as tmp:
        cmd = [exe, "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "step", "--", sys.executable, os.path.abspath(__file__),
               "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--headline-only"]
        try:
            run = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd="/tmp",
                                 env=dict(os.environ, TMPDIR="/tmp", PYTHONPATH=ROOT))
        except subprocess.TimeoutExpired:
            return None
        rows = _csv_rows(tmp, "kernel_trace.csv")
        durs = [(float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3 for r in rows
                if "self_chain_groups_kernel<5>" in r.get("Kernel_Name", "")]
        total = sum(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in rows) / 1e3
        if run.returncode != 0 or not durs:
            log("rocprofv3 --kernel-trace child failed (rc %d): %s" % (run.returncode, run.stderr[-300:]))
            return None
        return {"us_per_launch": round(sum(durs) / len(durs), 2), "launches": len(durs), "kernel_time_total_us": round(total, 1)}



def log(msg):
    print("[bench %.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)     # ~1.4 s timed region
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true",
                    help="skip the variant rates (eager, distinct images, last layer, trimmed): what the rocprofv3 "
                         "kernel-split run uses, so that the trace holds headline steps only")
    ap.add_argument("--no-config-legs", action="store_true",
                    help="skip the cfg 1 / 3 / 4 / 5 legs (tools/bench_legs.py) reported under \"configs\"")
    ap.add_argument("--legs", default=None, help="comma-separated subset of the config legs, e.g. cfg3,cfg5")
    ap.add_argument("--cpu-baseline-worker", nargs=2, type=int, metavar=("BATCH", "REPS"), help=argparse.SUPPRESS)
    ap.add_argument("--cpu-leg-worker", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--chain-worker", nargs=2, type=int, metavar=("BATCH", "LAUNCHES"), help=argparse.SUPPRESS)
    ap.add_argument("--no-profile-children", action="store_true",
                    help="do not run the rocprofv3 child processes (PMC traffic / in-step kernel trace); the line then quotes the "
                         "committed profiles/ files and says so")
    ap.add_argument("--workload", default="cfg2", choices=("cfg2", "cfg5"),
                    help="cfg2 (default, BASELINE.json's metric configuration) or the optional cfg-5 shape (ViT-L/14@336 bf16 body)")
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        cpu_baseline_worker(*args.cpu_baseline_worker)
        return
    if args.cpu_leg_worker:
        cpu_leg_worker(args.cpu_leg_worker)
        return
    if args.chain_worker:
        chain_worker(*args.chain_worker)
        return
    if args.workload == "cfg5":
        if args.steps == 100:
            args.steps = 5          # 120 ms per step
        main_cfg5(args)
        return

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # Test hooks for a 1-GPU box (the N > 1 path cannot otherwise be exercised there): MMX_BENCH_SHARE_DEVICE=1 puts every
    # rank on cuda:0, MMX_BENCH_BACKEND=gloo swaps RCCL (which refuses two ranks per GPU) for gloo.  Never set by the driver.
    backend = os.environ.get("MMX_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("MMX_BENCH_SHARE_DEVICE") else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # MMX_BENCH_FORCE_DIST=1 (tests/test_gpu_multigpu.py::test_rccl_world_size_one_smoke): join a WORLD-SIZE-1 RCCL group and run the
    # exchange step anyway, so that init_process_group("nccl", device_id=...) and the packed all_gather_into_tensor of this file
    # execute on the 1-GPU boxes the suite runs on.  Never set by the driver: a plain N = 1 run has no collective in its step.
    dist_on = world > 1 or bool(os.environ.get("MMX_BENCH_FORCE_DIST"))
    dist = None
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from transformer_mm_explainability_amd import clip_explainability as ce
    from transformer_mm_explainability_amd import clip_model, ops

    if os.environ.get("MMX_CHAIN_NT"):          # experiment hook (tools/gpu_r04.sh chain): cache policy of the chain's slab loads
        ops.set_option("self_chain_nt", int(os.environ["MMX_CHAIN_NT"]))
    log("imports done; building %s" % MODEL)
    model = clip_model.random_init(MODEL, seed=0)
    model = model.to(device)
    image, texts = synthetic_inputs(BATCH, device, seed=rank)
    # the exchange step moves the FULL per-sample result: image relevancy [49] and text relevancy [77 x 77] per pair
    row = 49 + 77 * 77
    gathered = torch.empty(world * BATCH, row, device=device) if dist_on else None
    packed = torch.empty(BATCH, row, device=device) if dist_on else None

    # The step is ~600 short launches; eager it is bound by the Python + launch path on the host, so the whole step
    # (forward, autograd backward, hand-written image backward, both chain launches) is captured ONCE into a hipGraph
    # and replayed.  Inputs are copied into the captured buffers each step (same values here: synthetic data).
    run = ce.GraphedInterpret(model, image, texts, start_layer=0, start_layer_text=0)

    def step():
        R_text, R_image = run(image, texts)
        if dist_on:   # the evaluators' exchange step: per-sample maps gathered on every rank (1.5 MB per rank)
            packed[:, :49] = R_image
            packed[:, 49:] = R_text.reshape(BATCH, -1)
            if backend == "nccl":
                dist.all_gather_into_tensor(gathered, packed)
            else:           # gloo test hook: list form
                dist.all_gather(list(gathered.view(world, BATCH, row).unbind(0)), packed)
        return R_text, R_image

    def sync():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    log("model on device; warmup")
    for _ in range(args.warmup):
        step()
    sync()
    log("timed region")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = world * BATCH / (elapsed / args.steps)

    log("timed region done: %.3f ms/step" % ms_per_step)

    def rate(fn, reps):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return BATCH / ((time.perf_counter() - t0) / reps)

    variants = None
    if not args.headline_only:
        reps = min(args.steps, 30)
        eager = rate(lambda: ce.interpret(image, texts, model, device, 0, 0), reps)
        # ---- variants, reported beside the headline (all eager unless noted)
        no_share = rate(lambda: ce.interpret(image, texts, model, device, 0, 0, share_image_forward=False), reps)
        # 64 DISTINCT images x 64 texts (nothing to share between the pairs), replayed from a hipGraph: the rate for a
        # workload that is not the reference's "one image, B captions" call
        g = torch.Generator().manual_seed(100 + rank)
        images64 = torch.randn(BATCH, 3, 224, 224, generator=g).to(device)
        run_distinct = ce.GraphedInterpret(model, images64, texts, 0, 0, share_image_forward=False)
        distinct_graph = rate(run_distinct, reps)
        del run_distinct, images64
        last_only = rate(lambda: ce.interpret(image, texts, model, device), reps)          # notebook default
        run_last = ce.GraphedInterpret(model, image, texts)
        last_only_graph = rate(run_last, reps)
        run_trim = ce.GraphedInterpret(model, image, texts, 0, 0, trim_text_padding=True)          # opt-in, exact
        trimmed_graph = rate(run_trim, reps)
        run_order = ce.GraphedInterpret(model, image, texts, 0, 0, image_chain_on_main=True)
        chain_on_main_graph = rate(run_order, reps)
        del run_last, run_trim, run_order
        variants = {"eager_maps_per_s": round(eager, 2),
                    "eager_B_image_copies_maps_per_s": round(no_share, 2),
                    "distinct_images_hipgraph_maps_per_s": round(distinct_graph, 2),
                    "last_layer_only_maps_per_s": {"eager": round(last_only, 2), "hipgraph": round(last_only_graph, 2)},
                    "trim_text_padding_hipgraph_maps_per_s": round(trimmed_graph, 2),
                    "image_chain_on_main_stream_hipgraph_maps_per_s": round(chain_on_main_graph, 2)}
        # the reference's OWN half-precision mode (convert_weights + the notebook's fp16 R chain): fp16
        # GEMMs, fp16 chain roundings, attention on the exact-fp32 kernels.  Reported for scale only -- the headline stays fp32.
        try:
            model.set_body_dtype(torch.float16)
            run_half = ce.GraphedInterpret(model, image, texts, 0, 0)
            variants["fp16_mode_hipgraph_maps_per_s"] = round(rate(run_half, reps), 2)
            del run_half
        except Exception as exc:                                               # a variant must not take the line down
            variants["fp16_mode_hipgraph_maps_per_s"] = None
            log("fp16 variant failed: %r" % (exc,))
        finally:
            model.set_body_dtype(torch.float32)
            ce.interpret(image, texts, model, device, 0, 0)                    # the fp32 slabs back in place for what follows
            torch.cuda.synchronize()

    roofline = None
    if not args.headline_only:
        # ---- roofline of the chain kernel, HIP events on the launch stream, buffers as the last step left them
        vis, txt = model.visual.transformer, model.transformer
        stream = torch.cuda.current_stream()

        ce.interpret(image, texts, model, device, 0, 0, share_image_forward=False)   # per-sample slabs for both towers

        # The text tower is causally masked: `interpret` launches its chain with MMX_CHAIN_CAUSAL (four-element chunks entirely above
        # the diagonal, where the probabilities are exact zeros, are not requested; same bits) -- that launch is the one timed; the
        # full read of the same slabs is timed beside it ("full_read").  `bytes_per_launch` stays SURVEY 8(d)'s algorithmic figure
        # (the whole slabs), `traffic` is what the counters saw.
        def chain(tr, causal=None):
            b = tr.buffers   # prepared launch: the timed loop is one C call per launch, not Python tensor plumbing
            return ops.ChainPlan([b.probs[l] for l in range(tr.layers)], [b.grads[l] for l in range(tr.layers)], BATCH,
                                 causal=(tr is txt) if causal is None else causal).launch

        def chain_bytes(tr, n):
            return 2 * tr.layers * BATCH * tr.heads * n * n * 4 + BATCH * n * n * 4

        log("kernel-only timing")
        # Each tower's launch is timed over ROTATING slab sets (text 3 x 293 MB, image 4 x 185 MB: more than the 256 MiB
        # Infinity Cache between two uses of a set), i.e. every byte comes from HBM as it does for slabs a backward pass has
        # just written; "same_buffers" (back-to-back launches over ONE set, what rounds 1-3 reported) is kept beside it.
        def rotating(tr, sets, keep=None, causal=None):
            b = tr.buffers
            causal = (tr is txt) if causal is None else causal
            plans = [chain(tr, causal)]
            if keep is None:
                keep = [([b.probs[l].clone() for l in range(tr.layers)], [b.grads[l].clone() for l in range(tr.layers)])
                        for _ in range(sets - 1)]
            for pr, gr in keep:
                plans.append(ops.ChainPlan(pr, gr, BATCH, causal=causal).launch)
            state = {"i": 0}

            def fn():
                plans[state["i"] % sets]()
                state["i"] += 1
            return fn, keep

        rot_txt, keep_txt = rotating(txt, 3)
        rot_img, keep_img = rotating(vis, 4)
        rot_txt_full, _ = rotating(txt, 3, keep_txt, causal=False)
        us_txt = kernel_time_us(rot_txt, 21, stream)
        us_txt_full = kernel_time_us(rot_txt_full, 21, stream)
        us_img = kernel_time_us(rot_img, 20, stream)
        us_txt_same = kernel_time_us(chain(txt), 20, stream)
        us_img_same = kernel_time_us(chain(vis), 20, stream)
        del keep_txt, keep_img
        by_txt, by_img = chain_bytes(txt, 77), chain_bytes(vis, 50)
        ach = by_txt / us_txt / 1e3  # GB/s
        # The same kernel at larger batches (rotating synthetic slab sets > the 256 MiB Infinity Cache): what the stream waves reach
        # once every CU has a workgroup of its own for the whole launch (B = 64 is the headline's batch and the line's `frac`)
        by_batch = {}
        for bb in (64, 128, 256):
            sets = 3 if bb == 64 else 2
            keep = [synthetic_text_slabs(bb, device, seed=10 + k) for k in range(sets)]
            bytes_b = 2 * 12 * bb * 8 * 77 * 77 * 4 + bb * 77 * 77 * 4
            us_by = {}
            for causal in (True, False):
                plans = [ops.ChainPlan(a_, g_, bb, causal=causal).launch for a_, g_ in keep]
                state = {"i": 0}

                def rot_fn():
                    plans[state["i"] % sets]()
                    state["i"] += 1
                us_by[causal] = kernel_time_us(rot_fn, 4 * sets + 1, stream)
            us_b = us_by[True]
            by_batch[str(bb)] = {"us_per_launch": round(us_b, 2), "achieved": round(bytes_b / us_b / 1e3, 1),
                                 "frac": round(bytes_b / us_b / 1e3 / HBM_PEAK_GBS, 4), "bytes_per_launch": bytes_b,
                                 "slab_sets": sets,
                                 "full_read": {"us_per_launch": round(us_by[False], 2),
                                               "frac": round(bytes_b / us_by[False] / 1e3 / HBM_PEAK_GBS, 4)}}
            del keep, plans
        torch.cuda.empty_cache()
        # HBM traffic of the same launch: measured in this run by two rocprofv3 --pmc child processes when rocprofv3 is on the box
        # (the counters cannot be read inside this process); otherwise the newest committed profiles/rNN_pmc_chain.json
        # (tools/pmc_chain_json.py; tests/test_profiles.py checks that it matches the committed PMC summaries) -- and the line says which
        traffic, traffic_source, traffic_split = None, None, None
        profile_children = not args.no_profile_children and world == 1
        if profile_children:
            log("rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE)")
            measured = measure_chain_counters()
            if measured:
                traffic = measured["fetch_bytes"] + measured["write_bytes"]
                traffic_split = {"fetch_bytes": measured["fetch_bytes"], "write_bytes": measured["write_bytes"]}
                traffic_source = measured["how"] + ", bytes per launch"
        if traffic is None:
            for pmc_file in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_chain.json")), reverse=True):
                pmc = json.load(open(pmc_file))
                pmc = pmc.get("self_chain_groups_kernel<5>") or pmc.get("self_chain_fused_kernel<5, 0>")
                if pmc:
                    traffic = pmc["fetch_bytes"] + pmc["write_bytes"]
                    traffic_source = ("committed PMC file %s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE x2 per the "
                                      "gfx950 correction; regenerated by tools/pmc_chain_json.py; NOT measured in this run%s), bytes per launch"
                                      % (os.path.relpath(pmc_file, ROOT), "" if not profile_children else ": the rocprofv3 child passes failed"))
                    break
        # the same kernel INSIDE the replayed step (beside the other tower's GEMMs), from the newest committed rocprofv3 kernel
        # trace of `bench.py --headline-only` (tools/gpu_round.sh): a committed file, not a measurement of this run
        in_step = None
        if profile_children:
            log("rocprofv3 --kernel-trace child (headline steps)")
            got = measure_chain_in_step()
            if got:
                in_step = {"us_per_launch": got["us_per_launch"], "achieved": round(by_txt / got["us_per_launch"] / 1e3, 1),
                           "frac": round(by_txt / got["us_per_launch"] / 1e3 / HBM_PEAK_GBS, 4), "launches": got["launches"],
                           "source": "measured in this run: child `bench.py --headline-only --steps 20` under rocprofv3 --kernel-trace, "
                                     "average duration of the text tower's chain launch inside the replayed steps"}
        for stats in ([] if in_step else sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_kernel_stats.txt")), reverse=True)):
            lines = open(stats).read().splitlines()
            # the text tower's launch: self_chain_groups_kernel<5> (round 5 on), self_chain_fused_kernel<5, 0, ...> in older summaries
            for ln in [x for x in lines if "self_chain_groups_kernel<5>" in x] + [x for x in lines if "self_chain_fused_kernel<5, 0" in x]:
                f = ln.split()
                avg = float(f[-4])
                in_step = {"us_per_launch": avg, "achieved": round(by_txt / avg / 1e3, 1), "frac": round(by_txt / avg / 1e3 / HBM_PEAK_GBS, 4),
                           "source": os.path.relpath(stats, ROOT) + " (committed rocprofv3 --kernel-trace --stats summary, avg_us column)"}
                break
            if in_step:
                break
        roofline = {"bound": "hbm", "kernel": "self_chain_groups_kernel<NT=5> (text tower, fp32 slabs)",
                    "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "timing": "HIP events on the launch stream over 21 stand-alone launches rotating over 3 slab sets (879 MB)",
                    "traffic": traffic, "traffic_source": traffic_source, "traffic_split": traffic_split,
                    # the same launch priced on the bytes the HBM counters saw instead of the algorithmic ones (the causal form reads less)
                    "on_measured_traffic": ({"achieved": round(traffic / us_txt / 1e3, 1), "frac": round(traffic / us_txt / 1e3 / HBM_PEAK_GBS, 4)}
                                            if traffic else None),
                    "by_batch": by_batch,
                    "bytes_per_launch": by_txt, "us_per_launch": round(us_txt, 2),
                    "causal_skip": {"what": "the text tower is causally masked (probabilities exactly 0 above the diagonal): the launch "
                                            "is made with MMX_CHAIN_CAUSAL and does not request the 16-byte chunks that lie entirely above "
                                            "the diagonal -- same bits as the full read (tests/test_gpu_ops.py).  `achieved` / `frac` price "
                                            "the launch on SURVEY 8(d)'s algorithmic bytes (the whole slabs); `traffic` is what the HBM "
                                            "counters saw for it; `full_read` is the same kernel over the same slabs without the flag",
                                    "requested_bytes_per_launch": causal_requested_bytes(txt.layers, BATCH, txt.heads, 77),
                                    "full_read": {"us_per_launch": round(us_txt_full, 2), "achieved": round(by_txt / us_txt_full / 1e3, 1),
                                                  "frac": round(by_txt / us_txt_full / 1e3 / HBM_PEAK_GBS, 4)}},
                    "same_buffers": {"us_per_launch": round(us_txt_same, 2), "achieved": round(by_txt / us_txt_same / 1e3, 1),
                                     "note": "20 back-to-back launches over ONE slab set: partly served by the Infinity Cache"},
                    "in_step": in_step,
                    "image_tower": {"kernel": "self_chain_groups_kernel<NT=4>", "bytes_per_launch": by_img,
                                    "us_per_launch": round(us_img, 2), "achieved": round(by_img / us_img / 1e3, 1),
                                    "same_buffers_us_per_launch": round(us_img_same, 2)},
                    "kernel_only_maps_per_s": round(BATCH / ((us_txt + us_img) * 1e-6), 1)}

    fl = step_flops(BATCH)
    tf = world * fl["total"] / (elapsed / args.steps) / 1e12
    split_file = None
    for cand in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_kernel_split.json")), reverse=True):
        split_file = cand
        break
    roofline_step = {"bound": "mfma", "achieved": round(tf / world, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(tf / world / FP32_MFMA_PEAK_TFLOPS, 4), "per": "GPU",
                     "flop_per_step": fl, "what": "exact-fp32 matrix FLOPs the step executes (body GEMMs forward + "
                     "input-gradient GEMMs, attention products, chain) / ms_per_step; the reference's fp32 contract rules "
                     "out the bf16 MFMA for the body",
                     "floor_ms_at_peak": round(fl["total"] / (FP32_MFMA_PEAK_TFLOPS * 1e12) * 1e3, 3),
                     "kernel_time_split": json.load(open(split_file)) if split_file else None,
                     "kernel_time_split_source": os.path.relpath(split_file, ROOT) if split_file else None}

    configs = None
    if world == 1 and not args.headline_only and not args.no_config_legs:
        # the other BASELINE.json configurations, each a bounded leg of its own (tools/bench_legs.py); the headline's graph
        # and slabs are released first (cfg 5 wants ~12 GB)
        import gc
        from tools import bench_legs
        del run
        for tr in (model.visual.transformer, model.transformer):
            tr.buffers = None
        model = None
        gc.collect()
        torch.cuda.empty_cache()
        log("config legs (cfg 1 / 3 / 4 / 5)")
        configs = bench_legs.run_all(kernel_time_us, log, only=args.legs.split(",") if args.legs else None)
        if not args.no_cpu_baseline:
            for name, leg in configs.items():       # the reference algorithm of each configuration on this box's host cores
                if "error" not in leg:
                    log("cpu baseline of %s (child process)" % name)
                    leg["cpu_baseline"] = cpu_leg(name, 180 if name == "cfg5" else 90)

    if rank == 0:
        line = {
            "metric": "relevancy maps/sec (fwd+bwd+rollout), CLIP ViT-B/32", "value": round(value, 2),
            "unit": "maps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "CLIP ViT-B/32 image<->text relevancy, batch=64 fp32 per GPU, all 12+12 layers "
                                   "(start_layer=0); random-init weights, synthetic image + token ids",
                       "global_batch": world * BATCH, "parallelism": "dp%d (independent batches, all-gather of maps)" % world,
                       **({"forced_world1_collective": "%s: world-size-1 group, the exchange step ran inside every timed step" % backend}
                          if dist_on and world == 1 else {}),
                       "launch": "whole step captured once into a hipGraph and replayed",
                       "image_tower": "forward shared by the batch (the reference API repeats ONE image B times), "
                                      "backward per sample",
                       # the headline shares ONE image-tower forward between the 64 captions (the reference's call shape); a batch of
                       # 64 DISTINCT images runs at this rate (hipGraph replay) -- the number that must travel with the headline
                       "distinct_images_maps_per_s": (variants or {}).get("distinct_images_hipgraph_maps_per_s"),
                       "variants": variants},
            "roofline": roofline,
            "roofline_step": roofline_step,
            "configs": configs,
        }
        if world == 1 and not args.no_cpu_baseline:
            log("cpu baseline (child process, <= 240 s)")
            line["cpu_baseline"] = cpu_baseline()
            log("cpu baseline done")
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
