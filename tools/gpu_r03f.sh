#!/bin/bash
TAG=${1:-r03f}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 1200 python -m pytest tests -q -m gpu --tb=line 2>&1 | tail -25 | tee $OUT/pytest_gpu.txt
cp gpurun_out/parity_errors.json $OUT/parity_errors.json 2>/dev/null
timeout 300 python tools/probe_attn_v3.py 128 0,2 2>&1 | grep -v amdgpu.ids | tee $OUT/attn_v3_probe.txt
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.log; python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step']); print(json.dumps(d['config']['variants'])); 
for k,v in d['configs'].items(): print(k, json.dumps(v)[:700])"
