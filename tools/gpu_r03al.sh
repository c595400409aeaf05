#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03al; mkdir -p $OUT
timeout 500 python -m pytest tests/test_gpu_multigpu.py -q -m gpu -x 2>&1 | tail -25 | tee $OUT/pytest.txt
