#!/bin/bash
# Round-5 GPU-box steps, one target per gpurun call:  tools/gpu_r05.sh <target> [tag]     outputs under gpurun_out/<tag>/
T=${1:?target}; TAG=${2:-$T}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT; nproc > $OUT/nproc.txt
case $T in
split)     # N > 128 chain route: the new product kernel, the overlap of reductions and products
  timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "bmm_tiles or split_and_rule7 or matmul" 2>&1 | tail -8 | tee $OUT/pytest_split.txt
  timeout 300 python tools/probe_chain_split_timing.py 2>&1 | grep -v amdgpu.ids | tee $OUT/chain_split_timing.txt
  ;;
bmmpmc)    # SQ counters of the tiles product kernel
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc -o bmm -- python tools/probe_bmm_only.py > /dev/null 2> $OUT/pmc.log
  python tools/pmc_sq.py $OUT/pmc/bmm_counter_collection.csv bmm_f32 > $OUT/bmm_sq.txt 2>&1; cat $OUT/bmm_sq.txt
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc2 -o bmm -- python tools/probe_bmm_only.py > /dev/null 2> $OUT/pmc2.log
  python tools/pmc_sq.py $OUT/pmc2/bmm_counter_collection.csv bmm_f32 > $OUT/bmm_sq2.txt 2>&1; cat $OUT/bmm_sq2.txt
  grep -i "bmm_f32" $OUT/pmc2/bmm_kernel_trace.csv | head -3
  rm -rf $OUT/pmc $OUT/pmc2
  ;;
groups)    # layer-group chain kernel with barrier-free stream waves: parity / bit-identity, then timing variants
  timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "layer_groups" 2>&1 | tail -15 | tee $OUT/pytest_groups.txt
  timeout 400 python tools/probe_chain_kernels.py 2>&1 | grep -v amdgpu.ids | grep -v "relay\|cols" | tee $OUT/chain_groups.txt
  ;;
cols)      # column-split chain kernel: bit identity, timing variants
  timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "strict_order_bit_identical" 2>&1 | tail -15 | tee $OUT/pytest_cols.txt
  timeout 400 python tools/probe_chain_kernels.py 2>&1 | grep -v amdgpu.ids | grep -v "relay\|fused kernel G=[23]\|groups kernel G=[24]" | tee $OUT/chain_cols.txt
  ;;
v3)        # cfg-5 attention backward pair: parity gate + timing
  timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity_fullsize.py -q -k "third_generation or bf16_backward or cfg5_bf16" 2>&1 | tail -5 | tee $OUT/pytest_v3.txt
  timeout 300 python tools/probe_attn_v3.py 128 0,2 2>&1 | grep -v amdgpu.ids | tee $OUT/attn_v3_probe.txt
  ;;
bmm)       # tile shape x prefetch depth of the large exact-fp32 product
  timeout 300 python -m pytest tests/test_gpu_ops.py -q -k "bmm_tiles or split_and_rule7 or matmul" 2>&1 | tail -5 | tee $OUT/pytest_bmm.txt
  BMM_VARIANTS=1 timeout 400 python tools/probe_chain_split_timing.py 2>&1 | grep -v amdgpu.ids | grep -v "chain\|avg_heads" | tee $OUT/bmm_variants.txt
  ;;
colsB)     # large batches: one workgroup per sample -- fused kernel vs the column kernel with C = 1
  for B in 128 160 256; do
    timeout 300 python tools/probe_chain_kernels.py $B 2>&1 | grep -v amdgpu.ids | grep "default\|per-sample\|cols C=1\|cols auto\|cols C=2\|groups kernel G=2" | tee -a $OUT/chain_cols_B.txt
  done
  ;;
groupsB)   # the same at other batch sizes (auto rule for the number of layer groups)
  for B in 16 32 96 128; do
    timeout 300 python tools/probe_chain_kernels.py $B 2>&1 | grep -v amdgpu.ids | grep -v "relay\|debug\|nt=0" | tee -a $OUT/chain_groups_B.txt
  done
  ;;
final)     # late additions: new chain tests, then the CPU thread sweep behind CPU_THREADS_CAP (host cores of the GPU box)
  timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "chain" 2>&1 | tail -4 | tee $OUT/pytest_chain.txt
  for t in 8 16 32 64 128; do timeout 120 python tools/probe_cpu_baseline.py $t 8 2>&1 | grep -v amdgpu.ids | tee -a $OUT/cpu_threads.txt; done
  ;;
suite)     # the whole GPU suite (after the pruning / parity-bound changes) + its timing
  timeout 1500 python -m pytest tests -q -m gpu --durations=15 2>&1 | tail -45 | tee $OUT/pytest_gpu.txt
  cp gpurun_out/parity_errors.json $OUT/parity.json 2>/dev/null
  ;;
*) echo "unknown target $T"; exit 2;;
esac
