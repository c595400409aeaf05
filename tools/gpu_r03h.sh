#!/bin/bash
TAG=${1:-r03h}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k detr_decoder_rows --tb=short 2>&1 | tail -15 | tee $OUT/pytest_detr_rows.txt
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/trace_cfg5 -o cfg5 -- python tools/probe_cfg5_trace.py 128 3 > /dev/null 2> $OUT/trace_cfg5.log
python tools/prof_summary.py $OUT/trace_cfg5/cfg5_results.db "" 2>&1 | head -45 | cut -c1-190 | tee $OUT/cfg5_kernels.txt
rm -rf $OUT/trace_cfg5
