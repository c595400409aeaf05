#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03ar; mkdir -p $OUT
timeout 120 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "linear_stream or small_linear or mfma_matmul" 2>&1 | tail -15 | tee $OUT/pytest.txt
timeout 120 python tools/probe_linear_stream.py 2>&1 | grep -v amdgpu.ids | tee $OUT/probe.txt
