#!/bin/bash
# r03y (experiment, NOT kept: the "self_chain_wide" option it toggles existed only in that build -- profiles/r03_chain_wide_probe.txt):
# chain kernel with wide (all-heads) raw-buffer load batches: tests, probe wide vs narrow, bench headline + roofline
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03y; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_clip.py tests/test_gpu_generators.py -q -x -k "chain or clip or small_linear or lxmert or visualbert" 2>&1 | tail -5 | tee $OUT/pytest.txt
python - <<'P' 2>&1 | grep -v amdgpu.ids | tee $OUT/chain_wide_probe.txt
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
for name, L, B, H, N in (("txt", 12, 64, 8, 77), ("img", 12, 64, 12, 50), ("vit-b/16 x8", 12, 8, 12, 197 if False else 128)):
    attn = [torch.softmax(torch.randn(B * H, N, N, device="cuda"), -1) for _ in range(L)]
    grad = [torch.randn(B * H, N, N, device="cuda") * 0.05 for _ in range(L)]
    mb = 2 * L * B * H * N * N * 4 / 1e6
    for wide in (0, 1):
        ops.set_option("self_chain_wide", wide)
        t = us(lambda: ops.relevancy_self_chain(attn, grad, B))
        print("%-12s L=%d B=%d H=%d N=%d  wide=%d: %.1f us  %.3f TB/s (%.1f MB)" % (name, L, B, H, N, wide, t, mb / t, mb))
ops.set_option("self_chain_wide", 1)
P
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-config-legs > $OUT/bench.json 2> $OUT/bench.log; tail -2 $OUT/bench.log
python - <<'P'
import json
d=json.loads(open("gpurun_out/r03y/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], json.dumps(d["roofline"])[:600])
P
