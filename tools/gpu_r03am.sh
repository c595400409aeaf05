#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03am; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o pert -- python tools/probe_lxmert_pert_trace.py 32 5 2>&1 | grep -v amdgpu.ids | tail -3 | tee $OUT/run.txt
python tools/prof_summary.py $OUT/trace/pert_results.db "" 2>&1 | head -45 | cut -c1-200 | tee $OUT/kernels.txt
rm -rf $OUT/trace
