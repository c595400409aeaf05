#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03ao; mkdir -p $OUT
python tools/probe_lxmert_pert_trace.py 32 8 untuned 2>&1 | grep "image test" | tee $OUT/before.txt
python tools/probe_lxmert_pert_trace.py 32 8 2>&1 | grep "image test" | tee $OUT/after.txt
