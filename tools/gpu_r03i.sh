#!/bin/bash
TAG=${1:-r03i}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_perturbation.py tests/test_gpu_generators.py -q -m gpu --tb=short 2>&1 | tail -40 | tee $OUT/pytest_tape.txt
timeout 300 python tools/probe_lxmert_pert.py 2>&1 | grep -v amdgpu.ids | tail -8 | tee $OUT/lxmert_probe.txt
