#!/bin/bash
TAG=${1:-r03e}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity_fullsize.py tests/test_gpu_lrp.py -q -m gpu --tb=short -k "third_generation or cfg5 or lrp" 2>&1 | tail -12 | tee $OUT/pytest_v3.txt
timeout 300 python tools/probe_attn_v3.py 128 2>&1 | grep -v amdgpu.ids | tee $OUT/attn_v3_probe.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_v3 -o v3 -- python tools/probe_attn_v3.py 128 2 > /dev/null 2> $OUT/trace_v3.log
python tools/prof_summary.py $OUT/trace_v3/v3_results.db "" --by-grid 2>&1 | head -8 | cut -c1-200 | tee $OUT/attn_v3_kernels.txt
rm -rf $OUT/trace_v3
