#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03o
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_generators.py tests/test_gpu_parity_fullsize.py -q -m gpu --tb=short -k "detr or graph_capture" 2>&1 | tail -12 | tee $OUT/pytest.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/tr -o detr -- python tools/probe_detr_trace.py 5 10 > /dev/null 2> $OUT/tr.log
python tools/prof_summary.py $OUT/tr/detr_results.db "" 2>&1 | head -30 | cut -c1-180 | tee $OUT/detr_kernels.txt
rm -rf $OUT/tr
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --legs cfg3 > $OUT/bench_cfg3.json 2> $OUT/bench.log; python -c "
import json; d=json.load(open('$OUT/bench_cfg3.json')); print(json.dumps(d['configs'])[:700])"
