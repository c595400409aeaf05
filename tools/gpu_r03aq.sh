#!/bin/bash
# ViT-B/16 one-image pass (cfg 1): tuned library-GEMM selection for its 197- / 1576-row shapes; leg before / after
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03aq; mkdir -p $OUT
CSV=transformer-mm-explainability_amd/tuning/tunableop_gfx950_vit_b16.csv
PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=6 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=2 timeout 100 python tools/tune_gemms.py vit_b16 $OUT/tunableop_gfx950_vit_b16.csv 2>&1 | grep -v amdgpu.ids | tail -2 | tee $OUT/tune.txt
ls $OUT
[ -s $OUT/tunableop_gfx950_vit_b16.csv ] || { [ -s $OUT/tunableop_gfx950_vit_b160.csv ] && cp $OUT/tunableop_gfx950_vit_b160.csv $OUT/tunableop_gfx950_vit_b16.csv; }
timeout 150 python - <<'P' 2>&1 | grep -v amdgpu.ids | tee $OUT/legs.txt
import os, shutil, sys
sys.path.insert(0, ".")
import bench
from tools import bench_legs
import torch
before = bench_legs.leg_cfg1(bench.kernel_time_us)
print("before:", {k: before[k] for k in ("rate", "ms", "multi_target")})
shutil.copy("gpurun_out/r03aq/tunableop_gfx950_vit_b16.csv", "transformer-mm-explainability_amd/tuning/tunableop_gfx950_vit_b16.csv")
after = bench_legs.leg_cfg1(bench.kernel_time_us)
print("after :", {k: after[k] for k in ("rate", "ms", "multi_target")})
P
