#!/bin/bash
# Round-4 GPU-box steps, one target per gpurun call:  tools/gpu_r04.sh <target> [tag]
#   outputs under gpurun_out/<tag>/ (default tag = target)
T=${1:?target}; TAG=${2:-$T}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT; nproc > $OUT/nproc.txt
case $T in
parity_bodies)   # the measured cfg-3 / cfg-4 legs + VisualBERT vs the independent oracle bodies
  timeout 900 python -m pytest tests/test_gpu_parity_bodies.py -q -x 2>&1 | tail -30 | tee $OUT/pytest.txt
  cp gpurun_out/parity_errors.json $OUT/parity.json 2>/dev/null
  timeout 300 python tools/probe_chain_nt.py 2>&1 | grep -v amdgpu.ids | tee $OUT/chain_nt.txt
  ;;
schedule)        # two-phase bi-modal schedule kernel: parity tests, timing, and the LRP route's cost (before fusing it)
  timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_generators.py tests/test_gpu_perturbation.py tests/test_gpu_lrp.py -q -x -k "lxmert or schedule or tapes or diag_word" 2>&1 | tail -15 | tee $OUT/pytest.txt
  timeout 300 python tools/probe_schedule.py 2>&1 | grep -v amdgpu.ids | tee $OUT/schedule.txt
  timeout 300 python tools/probe_lrp.py both 5 2>&1 | grep -v amdgpu.ids | tee $OUT/lrp.txt
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_lrp -o lrp -- python tools/probe_lrp.py both 3 > /dev/null 2> $OUT/trace_lrp.log
  python tools/prof_summary.py $OUT/trace_lrp/lrp_results.db "" 2>&1 | head -45 | cut -c1-200 > $OUT/lrp_kernels.txt; head -30 $OUT/lrp_kernels.txt
  rm -rf $OUT/trace_lrp
  ;;
schedule2)       # schedule kernel only: tests + timing with the phase-skip flags
  timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_generators.py tests/test_gpu_perturbation.py tests/test_gpu_lrp.py -q -k "lxmert or schedule or tapes or diag_word" 2>&1 | tail -15 | tee $OUT/pytest.txt
  timeout 300 python tools/probe_schedule.py 2>&1 | grep -v amdgpu.ids | tee $OUT/schedule.txt
  ;;
lrp)             # fused LRP rule kernels: parity (all LRP suites) + the pass times
  timeout 900 python -m pytest tests/test_gpu_lrp.py tests/test_gpu_lrp_route.py -q 2>&1 | tail -15 | tee $OUT/pytest.txt
  timeout 300 python tools/probe_lrp.py both 5 2>&1 | grep -v amdgpu.ids | tee $OUT/lrp.txt
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_lrp -o lrp -- python tools/probe_lrp.py both 3 > /dev/null 2> $OUT/trace_lrp.log
  python tools/prof_summary.py $OUT/trace_lrp/lrp_results.db "" 2>&1 | head -45 | cut -c1-200 > $OUT/lrp_kernels.txt; head -30 $OUT/lrp_kernels.txt
  rm -rf $OUT/trace_lrp
  ;;
chain)           # chain kernel: SQ / TCC counters over rotating slabs, nt vs default policy inside the replayed step
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/pmc_sq -o chain -- python tools/probe_chain_nt.py > $OUT/probe_under_pmc.txt 2> $OUT/pmc_sq.log
  python tools/pmc_sq.py $OUT/pmc_sq/chain_counter_collection.csv self_chain_fused > $OUT/chain_sq.txt 2>&1; cat $OUT/chain_sq.txt
  timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc -o chain -- python tools/probe_chain_nt.py > /dev/null 2> $OUT/pmc_tcc.log || tail -3 $OUT/pmc_tcc.log
  python tools/pmc_sq.py $OUT/pmc_tcc/chain_counter_collection.csv self_chain_fused > $OUT/chain_tcc.txt 2>&1; cat $OUT/chain_tcc.txt
  for NT in 0 1; do
    MMX_CHAIN_NT=$NT timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_nt$NT -o bench -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --headline-only > $OUT/bench_nt$NT.json 2> $OUT/trace_nt$NT.log
    python tools/prof_summary.py $OUT/trace_nt$NT/bench_results.db "self_chain" --by-grid > $OUT/chain_in_step_nt$NT.txt 2>&1; echo "nt=$NT"; cat $OUT/chain_in_step_nt$NT.txt | cut -c1-200; cut -c1-160 $OUT/bench_nt$NT.json
  done
  rm -rf $OUT/pmc_sq $OUT/pmc_tcc $OUT/trace_nt0 $OUT/trace_nt1
  ;;
cfg5)            # bf16 body: parity suites + step time + launch-count audit (kernel trace sorted by CALLS)
  timeout 900 python -m pytest tests/test_gpu_parity_fullsize.py tests/test_gpu_clip.py tests/test_gpu_ops.py -q -k "cfg5 or bf16" 2>&1 | tail -8 | tee $OUT/pytest.txt
  timeout 300 python -c "
import torch, time, sys
sys.path.insert(0, '.')
from tools import bench_legs
from transformer_mm_explainability_amd import clip_explainability as ce
model, image, texts, _, _ = bench_legs.cfg5_setup(128, torch.device('cuda'))
f = lambda: ce.interpret(image, texts, model, 'cuda', start_layer=0, start_layer_text=0)
for _ in range(2): f()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): f()
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
print('cfg5 step: %.2f ms = %.1f maps/s' % (ms, 128 / ms * 1e3))
" 2>&1 | grep -v amdgpu.ids | tee $OUT/step.txt
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/trace_cfg5 -o cfg5 -- python tools/probe_cfg5_trace.py 128 3 > /dev/null 2> $OUT/trace_cfg5.log
  python tools/prof_summary.py $OUT/trace_cfg5/cfg5_results.db "" 2>&1 | head -40 | cut -c1-190 > $OUT/cfg5_step_kernels.txt; head -24 $OUT/cfg5_step_kernels.txt
  python tools/prof_summary.py $OUT/trace_cfg5/cfg5_results.db "copy" 2>&1 | cut -c1-190 | tee $OUT/cfg5_copy_kernels.txt
  rm -rf $OUT/trace_cfg5
  ;;
attn5)           # cfg-5 attention backward pair stand-alone + equivalence tests + the step
  timeout 600 python -m pytest tests/test_gpu_parity_fullsize.py -q -k "third_generation or cfg5_bf16_body or bf16_error_by_depth" 2>&1 | tail -5 | tee $OUT/pytest.txt
  timeout 300 python tools/probe_attn_v3.py 128 2>&1 | grep -v amdgpu.ids | tee $OUT/attn_v3_probe.txt
  timeout 300 python -c "
import torch, time, sys
sys.path.insert(0, '.')
from tools import bench_legs
from transformer_mm_explainability_amd import clip_explainability as ce
model, image, texts, _, _ = bench_legs.cfg5_setup(128, torch.device('cuda'))
f = lambda: ce.interpret(image, texts, model, 'cuda', start_layer=0, start_layer_text=0)
for _ in range(2): f()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): f()
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
print('cfg5 step: %.2f ms = %.1f maps/s' % (ms, 128 / ms * 1e3))
" 2>&1 | grep -v amdgpu.ids | tee $OUT/step.txt
  ;;
counters2)       # second counter set (matrix-pipe busy, instruction counts, instruction fetch, vmem level) on the cfg-5 pair and the schedule
  C2="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_IFETCH SQ_INST_LEVEL_VMEM"
  timeout 300 rocprofv3 --pmc $C2 --kernel-trace --output-format csv -d $OUT/pmc_v3 -o v3 -- python tools/probe_attn_v3.py 16 2 > /dev/null 2> $OUT/pmc_v3.log
  python tools/pmc_sq.py $OUT/pmc_v3/v3_counter_collection.csv _v3_ > $OUT/attn_v3_counters2.txt 2>&1; cat $OUT/attn_v3_counters2.txt
  C3="SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_WAVES"
  timeout 300 rocprofv3 --pmc $C3 --kernel-trace --output-format csv -d $OUT/pmc_v3b -o v3 -- python tools/probe_attn_v3.py 16 2 > /dev/null 2> $OUT/pmc_v3b.log
  python tools/pmc_sq.py $OUT/pmc_v3b/v3_counter_collection.csv _v3_ > $OUT/attn_v3_counters3.txt 2>&1; cat $OUT/attn_v3_counters3.txt
  timeout 300 rocprofv3 --pmc $C2 --kernel-trace --output-format csv -d $OUT/pmc_sched -o s -- python tools/probe_schedule.py > /dev/null 2> $OUT/pmc_sched.log
  python tools/pmc_sq.py $OUT/pmc_sched/s_counter_collection.csv lxmert_schedule > $OUT/schedule_counters2.txt 2>&1; cat $OUT/schedule_counters2.txt
  rm -rf $OUT/pmc_v3 $OUT/pmc_v3b $OUT/pmc_sched
  ;;
counters3)       # memory-side counters of the cfg-5 pair (SQ LDS / VMEM view).  NOTE: the TCP / TA / TCC sets this target first carried
                 # (8 counters of ONE block per pass) made rocprofv3 abort with "Request exceeds the capabilities of the hardware to collect"
                 # and then sit until its timeout -- 15 GPU-minutes for nothing.  At most 4 counters of a block per pass, short timeouts.
  i=0
  for C in "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum"; do
    i=$((i+1))
    timeout 90 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$i -o v -- python tools/probe_attn_v3.py 16 2,3 > /dev/null 2> $OUT/pmc_$i.log
    python tools/pmc_sq.py $OUT/pmc_$i/v_counter_collection.csv attn_bwd_ > $OUT/attn_counters_mem_$i.txt 2>&1 || tail -5 $OUT/pmc_$i.log
    cat $OUT/attn_counters_mem_$i.txt
    rm -rf $OUT/pmc_$i
  done
  ;;
ablate)          # timing ablations of the cfg-5 pair (library built with -DMMX_ATTN_ABLATE; results of the ablated kernels are garbage)
  for A in ${ABLS:-0 2 3 4 7}; do
    MMX_ATTN_ABLATE=$A timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/abl_$A -o a -- python tools/probe_attn_v3.py 128 2,3 > $OUT/abl_$A.txt 2> $OUT/abl_$A.log
    echo "== ablation $A" | tee -a $OUT/ablate.txt
    grep -v amdgpu.ids $OUT/abl_$A.txt | tee -a $OUT/ablate.txt
    python - $OUT/abl_$A/a_kernel_stats.csv <<'PY' | tee -a $OUT/ablate.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "attn_bwd" in r["Name"] or "prep_" in r["Name"] or "rel_row" in r["Name"]:
        print("    %-70s calls %4s  avg %9.1f us" % (r["Name"].replace("void mmx::(anonymous namespace)::", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
    rm -rf $OUT/abl_$A
  done
  ;;
head)            # whole-head attention kernels: stagger sweep (co-resident workgroups out of phase) + the fp16-mode full-size test
  timeout 300 python -m pytest tests/test_gpu_parity_fullsize.py -q -s -k "fp16_mode" 2>&1 | grep -E "passed|failed|relative errors|Error" | tee $OUT/pytest.txt
  for S in 0 100 300 600 1200; do
    MMX_HEAD_STAGGER=$S timeout 120 python tools/probe_head_attention.py 2>&1 | grep -E "stagger|text|image" | grep -v checksums | tee -a $OUT/head_stagger.txt
  done
  ;;
*) echo "unknown target $T"; exit 2;;
esac
