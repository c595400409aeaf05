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
*) echo "unknown target $T"; exit 2;;
esac
