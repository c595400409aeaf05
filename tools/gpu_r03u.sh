#!/bin/bash
# r03u: small-grid split forward attention kernel, NT layout of the FFN backward GEMM, value projection beside q/k
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03u; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_generators.py tests/test_gpu_lrp.py tests/test_gpu_parity_fullsize.py -q -x -k "attn or detr or DETR or small_linear" 2>&1 | tail -8 | tee $OUT/pytest.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --legs cfg3 > $OUT/bench.json 2> $OUT/bench.log; tail -2 $OUT/bench.log
python - <<'P'
import json
d=json.loads(open("gpurun_out/r03u/bench.json").read().strip().splitlines()[-1])
for k,v in d["configs"].items(): print(k, {kk:vv for kk,vv in v.items() if kk in ("rate","ms","K10","K20")})
P
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_detr -o detr -- python tools/probe_detr_trace.py 5 10 rows > /dev/null 2> $OUT/trace_detr.log
python tools/prof_summary.py $OUT/trace_detr/detr_results.db "" 2>&1 | head -24 | cut -c1-190 > $OUT/detr_rows_kernels.txt; cat $OUT/detr_rows_kernels.txt
rm -rf $OUT/trace_detr
