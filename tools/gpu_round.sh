#!/bin/bash
# One GPU-box round: gpu tests, smoke, bench (JSON line), rocprofv3 kernel trace + PMC passes. Outputs under gpurun_out/$TAG.
#   $2 = "notest" skips the pytest / smoke legs (profiles only); "lite" keeps them and skips the cfg-5 / DETR side traces at the end
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
nproc > $OUT/nproc.txt
if [ "$2" != "notest" ]; then
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -12 | tee $OUT/pytest_gpu.txt
cp gpurun_out/parity_errors.json $OUT/parity.json 2>/dev/null
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
fi
timeout 900 python bench.py --steps 100 --warmup 5 > $OUT/bench.json 2> $OUT/bench.log; tail -2 $OUT/bench.log; cut -c1-400 $OUT/bench.json
# kernel trace of headline steps only (2 capture warm-ups + 3 warm-up replays + 20 timed replays = 25 steps in the trace)
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --headline-only > $OUT/bench_traced.json 2> $OUT/trace.log
python tools/prof_summary.py $OUT/trace/bench_results.db "" --by-grid > $OUT/kernel_stats.txt 2>&1; grep mmx $OUT/kernel_stats.txt | cut -c1-200
python tools/kernel_split.py $OUT/trace/bench_results.db 26 > $OUT/kernel_split.json; cat $OUT/kernel_split.json
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --headline-only > /dev/null 2> $OUT/pmc_fetch.log
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --headline-only > /dev/null 2> $OUT/pmc_write.log
python tools/pmc_summary.py $OUT/pmc_fetch/bench_counter_collection.csv mmx:: > $OUT/pmc_fetch.txt; python tools/pmc_summary.py $OUT/pmc_write/bench_counter_collection.csv mmx:: > $OUT/pmc_write.txt
cat $OUT/pmc_fetch.txt $OUT/pmc_write.txt
# trace of the DEFAULT command's legs (headline + variants + the stand-alone chain launches the roofline is measured on + the
# cfg 1 / 3 / 4 / 5 legs): the chain kernels' min / avg there include the 2 x 21 stand-alone launches
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_full -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/trace_full.log
python tools/prof_summary.py $OUT/trace_full/bench_results.db "self_chain" --by-grid > $OUT/chain_kernel_trace.txt 2>&1; cat $OUT/chain_kernel_trace.txt | cut -c1-200
python tools/prof_summary.py $OUT/trace_full/bench_results.db "" --by-grid 2>&1 | grep -E "kernel  |mmx" | head -70 | cut -c1-200 > $OUT/cfg_legs.txt
# round 4: the bi-modal schedule kernel (one workgroup per sample vs two-phase) and the LRP route's cost
if [ -z "$SKIP_PROBES" ]; then      # (SKIP_PROBES=1: a re-run late in a round, the probes' files are already committed)
timeout 200 python tools/probe_schedule.py 2>&1 | grep -v amdgpu.ids > $OUT/schedule_probe.txt; cat $OUT/schedule_probe.txt
timeout 200 python tools/probe_lrp.py both 5 2>&1 | grep -v amdgpu.ids > $OUT/lrp_probe.txt; cat $OUT/lrp_probe.txt
rocprofv3 --list-avail > $OUT/counters_avail.txt 2>&1
fi
if [ "$2" == "lite" ]; then rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/trace $OUT/trace_full; exit 0; fi
# cfg 5: the step's kernel split, the attention backward pair alone (generations 0 / 2 / 3 of the option attn_bf16_v3), SQ counters of the default pair
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/trace_cfg5 -o cfg5 -- python tools/probe_cfg5_trace.py 128 3 > /dev/null 2> $OUT/trace_cfg5.log
python tools/prof_summary.py $OUT/trace_cfg5/cfg5_results.db "" 2>&1 | head -30 | cut -c1-190 > $OUT/cfg5_step_kernels.txt
timeout 300 python tools/probe_attn_v3.py 128 0,2,3 3 2>&1 | grep -v amdgpu.ids > $OUT/attn_v3_probe.txt; cat $OUT/attn_v3_probe.txt
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/pmc_v3 -o v3 -- python tools/probe_attn_v3.py 16 3 > /dev/null 2> $OUT/pmc_v3.log
python tools/pmc_sq.py $OUT/pmc_v3/v3_counter_collection.csv attn_bwd_ > $OUT/attn_v3_sq.txt 2>&1; cat $OUT/attn_v3_sq.txt
if [ -z "$SKIP_PROBES" ]; then
# cfg 3: DETR K = 10 pass with the three-launch decoder rules (rows of R_q_i only)
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_detr -o detr -- python tools/probe_detr_trace.py 5 10 rows > /dev/null 2> $OUT/trace_detr.log
python tools/prof_summary.py $OUT/trace_detr/detr_results.db "" 2>&1 | head -40 | cut -c1-190 > $OUT/detr_rows_kernels.txt
fi
rm -rf $OUT/trace_detr $OUT/pmc_fetch $OUT/pmc_write $OUT/trace $OUT/trace_full $OUT/trace_cfg5 $OUT/pmc_v3
