#!/bin/bash
# Quick GPU-box check: the GPU test suite (with the parity-error record) + one default bench line.  Outputs under gpurun_out/$TAG.
TAG=${1:-r03q}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
nproc > $OUT/nproc.txt
timeout 1200 python -m pytest tests -q -m gpu -x --durations=15 2>&1 | tail -40 | tee $OUT/pytest_gpu.txt
cp gpurun_out/parity_errors.json $OUT/parity_errors.json 2>/dev/null
timeout 600 python bench.py --steps 50 --warmup 5 > $OUT/bench.json 2> $OUT/bench.log; tail -8 $OUT/bench.log; cut -c1-600 $OUT/bench.json
