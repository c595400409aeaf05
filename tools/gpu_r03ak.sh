#!/bin/bash
# whole-head attention kernels: transposed-output products (16-byte stores) + prefetched phase C, A / B against the previous build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03ak; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_clip.py tests/test_gpu_parity_fullsize.py -q -m gpu -x -k "attn or attention or cfg2 or clip or small_linear" 2>&1 | tail -4 | tee $OUT/pytest.txt
timeout 200 python tools/probe_head_attention.py tools/_ab/libmmx_hip_base.so 2>&1 | grep -v amdgpu.ids | tee $OUT/base.txt
timeout 200 python tools/probe_head_attention.py transformer-mm-explainability_amd/csrc/libmmx_hip.so noyard 2>&1 | grep -v amdgpu.ids | tee $OUT/new.txt
