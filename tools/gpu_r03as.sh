#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03as; mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/trace -o vit -- python tools/probe_vit_trace.py 20 2>&1 | grep "hipGraph replay" | tee $OUT/run.txt
python tools/prof_summary.py $OUT/trace/vit_results.db "" 2>&1 | head -60 | cut -c1-200 > $OUT/kernels.txt
rm -rf $OUT/trace
