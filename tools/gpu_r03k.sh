#!/bin/bash
TAG=${1:-r03k}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 1200 python -m pytest tests -q -m gpu --tb=line 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
timeout 300 python tools/probe_lxmert_pert.py 2>&1 | grep -v amdgpu.ids | tail -4 | tee $OUT/lxmert_probe.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/tr_split -o split -- python tools/probe_chain_split.py > $OUT/chain_split_shapes.txt 2> $OUT/tr_split.log
python tools/prof_summary.py $OUT/tr_split/split_results.db mmx --by-grid 2>&1 | head -12 | cut -c1-180 | tee $OUT/chain_split_kernels.txt
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_split -o split -- python tools/probe_chain_split.py > /dev/null 2> $OUT/pmc_split.log
python tools/pmc_summary.py $OUT/pmc_split/split_counter_collection.csv mmx:: 2>&1 | tee $OUT/chain_split_fetch.txt
rm -rf $OUT/tr_split $OUT/pmc_split
