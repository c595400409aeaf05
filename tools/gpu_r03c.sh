#!/bin/bash
TAG=${1:-r03c}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_lrp.py -q -m gpu --tb=short 2>&1 > $OUT/pytest_lrp.txt; tail -30 $OUT/pytest_lrp.txt
timeout 900 python -m pytest tests/test_gpu_parity_fullsize.py -q -m gpu -k "third_generation" --tb=short 2>&1 | tail -8 | tee $OUT/pytest_v3.txt
timeout 200 python tools/probe_gemm_split.py 2>&1 | grep -v amdgpu.ids | tee $OUT/gemm_split_probe.txt
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/pmc_v3 -o v3 -- python tools/probe_attn_v3.py 16 2 > /dev/null 2> $OUT/pmc_v3.log
python tools/pmc_sq.py $OUT/pmc_v3/v3_counter_collection.csv _v3_ 2>&1 | tee $OUT/attn_v3_sq.txt
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_v3b -o v3 -- python tools/probe_attn_v3.py 16 2 > /dev/null 2> $OUT/pmc_v3b.log
python tools/pmc_sq.py $OUT/pmc_v3b/v3_counter_collection.csv _v3_ 2>&1 | tee $OUT/attn_v3_sq2.txt
tail -3 $OUT/pmc_v3b.log
rm -rf $OUT/pmc_v3 $OUT/pmc_v3b
