#!/bin/bash
# r03 second GPU visit: LRP + v3 attention backward tests, v3 probe with a kernel trace, cfg5 leg.
TAG=${1:-r03b}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_lrp.py tests/test_gpu_lrp_route.py -q -m gpu 2>&1 | tail -40 > $OUT/pytest_lrp.txt; tail -25 $OUT/pytest_lrp.txt
timeout 900 python -m pytest tests/test_gpu_parity_fullsize.py -q -m gpu -k "third_generation or cfg5" 2>&1 | tail -30 > $OUT/pytest_v3.txt; tail -15 $OUT/pytest_v3.txt
timeout 300 python tools/probe_attn_v3.py 128 2>&1 | grep -v amdgpu.ids | tee $OUT/attn_v3_probe.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_v3 -o v3 -- python tools/probe_attn_v3.py 128 2 > /dev/null 2> $OUT/trace_v3.log
python tools/prof_summary.py $OUT/trace_v3/v3_results.db "" --by-grid 2>&1 | head -14 | cut -c1-200 | tee $OUT/attn_v3_kernels.txt
rm -rf $OUT/trace_v3
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --legs cfg5 > $OUT/bench_cfg5leg.json 2> $OUT/bench_cfg5leg.log; python -c "
import json; d=json.load(open('$OUT/bench_cfg5leg.json')); print(json.dumps(d['configs'])[:900])"
