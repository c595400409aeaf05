"""GPU box: which library GEMM shapes does ONE eager headline step (CLIP ViT-B/32, B = 64, shared image forward) issue?  Run under
ROCBLAS_LAYER=2 (rocBLAS bench-style log on stderr) and summarise (m, n, k, batch, transposes) with counts."""
import os, sys, re, subprocess, collections
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from transformer_mm_explainability_amd import clip_explainability as ce, clip_model
    model = clip_model.random_init(bench.MODEL, seed=0).cuda()
    image, texts = bench.synthetic_inputs(64, "cuda", 0)
    ce.interpret(image, texts, model, "cuda", 0, 0)
    torch.cuda.synchronize()
    print("MARK", file=sys.stderr, flush=True)
    ce.interpret(image, texts, model, "cuda", 0, 0)
    torch.cuda.synchronize()
    sys.exit(0)
env = dict(os.environ, ROCBLAS_LAYER="2", HIPBLASLT_LOG_LEVEL="0")
out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], capture_output=True, text=True, env=env)
log = out.stderr.split("MARK")[-1]
cnt = collections.Counter()
for ln in log.splitlines():
    if "rocblas-bench" not in ln:
        continue
    g = lambda k: (re.search(r"--%s (\S+)" % k, ln) or re.search(r"-%s (\S+)" % k, ln))
    f = re.search(r"-f (\S+)", ln).group(1)
    key = (f, g("transposeA").group(1), g("transposeB").group(1), g("m").group(1), g("n").group(1), g("k").group(1),
           (g("batch_count").group(1) if g("batch_count") else "1"))
    cnt[key] += 1
print("rocBLAS calls in one eager step: %d" % sum(cnt.values()))
for key, n in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print("%4d x %s" % (n, " ".join(key)))
if not cnt:
    print(out.stderr[-3000:])
