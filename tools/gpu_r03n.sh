#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03n
mkdir -p $OUT
for i in 1 2; do timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -4; done | tee $OUT/pytest_gpu_twice.txt
cp gpurun_out/parity_errors.json $OUT/parity.json
