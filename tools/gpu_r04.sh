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
*) echo "unknown target $T"; exit 2;;
esac
