#!/bin/bash
# Round-5 GPU-box steps, one target per gpurun call:  tools/gpu_r05.sh <target> [tag]     outputs under gpurun_out/<tag>/
T=${1:?target}; TAG=${2:-$T}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT; nproc > $OUT/nproc.txt
case $T in
relay)     # relay chain kernel: bit-identity test, stand-alone timing with variants, then the whole GPU suite (relative parity bound)
  timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "relay" 2>&1 | tail -15 | tee $OUT/pytest_relay.txt
  timeout 300 python tools/probe_chain_relay.py 2>&1 | grep -v amdgpu.ids | tee $OUT/chain_relay.txt
  timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -25 | tee $OUT/pytest_gpu.txt
  cp gpurun_out/parity_errors.json $OUT/parity.json 2>/dev/null
  ;;
relay2)    # relay kernel after the fixes: test, element-level diagnostic, timing variants
  timeout 300 python -m pytest tests/test_gpu_ops.py -q -k "relay" 2>&1 | tail -15 | tee $OUT/pytest_relay.txt
  timeout 120 python tools/probe_relay_debug.py 2>&1 | grep -v amdgpu.ids | tee $OUT/relay_debug.txt
  timeout 300 python tools/probe_chain_relay.py 2>&1 | grep -v amdgpu.ids | tee $OUT/chain_relay.txt
  timeout 600 python -m pytest tests/test_gpu_vit.py tests/test_gpu_clip.py -q -x 2>&1 | tail -8 | tee $OUT/pytest_vit_clip.txt
  ;;
*) echo "unknown target $T"; exit 2;;
esac
