#!/bin/bash
# r03aa: chain kernel with software-pipelined stream waves: tests, pipe on/off timing, bench headline + roofline
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03aa; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_clip.py tests/test_gpu_generators.py -q -x -k "chain or clip or lxmert or visualbert or rules" 2>&1 | tail -4 | tee $OUT/pytest.txt
python - <<'P' 2>&1 | grep -v amdgpu.ids | tee $OUT/chain_pipe_probe.txt
import sys, torch
sys.path.insert(0, ".")
from transformer_mm_explainability_amd import ops
def us(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for name, L, B, H, N in (("txt", 12, 64, 8, 77), ("img", 12, 64, 12, 50), ("lxmert-l", 9, 32, 12, 14), ("vit-b N=128", 12, 8, 12, 128)):
    attn = [torch.softmax(torch.randn(B * H, N, N, device="cuda"), -1) for _ in range(L)]
    grad = [torch.randn(B * H, N, N, device="cuda") * 0.05 for _ in range(L)]
    mb = 2 * L * B * H * N * N * 4 / 1e6
    for pipe in (0, 1, 2, 4):
        ops.set_option("self_chain_pipe", pipe)
        t = us(lambda: ops.relevancy_self_chain(attn, grad, B))
        print("%-12s L=%d B=%d H=%d N=%d  pipe=%d: %.1f us  %.3f TB/s (%.1f MB)" % (name, L, B, H, N, pipe, t, mb / t, mb))
ops.set_option("self_chain_pipe", 4)
P
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-config-legs > $OUT/bench.json 2> $OUT/bench.log; tail -2 $OUT/bench.log
python - <<'P'
import json
d=json.loads(open("gpurun_out/r03aa/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k: d["roofline"][k] for k in ("achieved", "frac", "us_per_launch") if k in d["roofline"]})
P
