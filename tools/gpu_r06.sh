#!/bin/bash
# Round-6 GPU-box steps, one target per gpurun call:  tools/gpu_r06.sh <target> [tag]     outputs under gpurun_out/<tag>/
T=${1:?target}; TAG=${2:-$T}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT; nproc > $OUT/nproc.txt
case $T in
suite)     # the whole GPU suite + smoke + a short headline bench
  timeout 2400 python -m pytest tests -q -m gpu --durations=20 -x 2>&1 | tail -60 | tee $OUT/pytest_gpu.txt
  cp gpurun_out/parity_errors.json $OUT/parity.json 2>/dev/null
  timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
  timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --headline-only > $OUT/bench_headline.json 2> $OUT/bench.log; cut -c1-600 $OUT/bench_headline.json
  ;;
suite_all) # the same without -x (see every failure)
  timeout 2400 python -m pytest tests -q -m gpu --durations=20 2>&1 | tail -80 | tee $OUT/pytest_gpu.txt
  cp gpurun_out/parity_errors.json $OUT/parity.json 2>/dev/null
  ;;
tests)     # selected tests:  TESTS="tests/test_x.py -k foo" tools/gpu_r06.sh tests
  timeout 1500 python -m pytest $TESTS -q 2>&1 | tail -40 | tee $OUT/pytest_sel.txt
  cp gpurun_out/parity_errors.json $OUT/parity.json 2>/dev/null
  ;;
cmd)       # arbitrary probe:  CMD="python tools/probe_x.py ..." tools/gpu_r06.sh cmd tag
  timeout ${CMD_TIMEOUT:-900} bash -c "$CMD" 2>&1 | grep -v amdgpu.ids | tail -${CMD_TAIL:-120} | tee $OUT/cmd.txt
  ;;
*) echo "unknown target $T"; exit 2;;
esac
