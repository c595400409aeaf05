#!/bin/bash
TAG=${1:-r03d}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 300 python tools/probe_attn_v3.py 128 2>&1 | grep -v amdgpu.ids | tee $OUT/attn_v3_probe.txt
timeout 1200 python -m pytest tests -q -m gpu -x --tb=short 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
cp gpurun_out/parity_errors.json $OUT/parity_errors.json 2>/dev/null
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --legs cfg5 > $OUT/bench_cfg5leg.json 2> $OUT/bench_cfg5leg.log; python -c "
import json; d=json.load(open('$OUT/bench_cfg5leg.json')); print(json.dumps(d['configs'])[:900])"
