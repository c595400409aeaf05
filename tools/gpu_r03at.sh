#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03at; mkdir -p $OUT
timeout 200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_vit.py -q -m gpu -x -k "avg_heads or vit" 2>&1 | tail -12 | tee $OUT/pytest.txt
timeout 100 python tools/probe_vit_trace.py 30 2>&1 | grep "hipGraph replay" | tee $OUT/run.txt
