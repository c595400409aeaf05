#!/bin/bash
# LXMERT perturbation re-runs: vectorised keep-mask builders + a tuned GEMM selection for the big-row forwards
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03an; mkdir -p $OUT
CSV=transformer-mm-explainability_amd/tuning/tunableop_gfx950_lxmert_pert.csv
python tools/probe_lxmert_pert_trace.py 32 5 untuned 2>&1 | grep "image test" | tee $OUT/before.txt
PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=6 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=2 timeout 330 python tools/tune_gemms.py lxmert_pert $OUT/tunableop_gfx950_lxmert_pert.csv 2>&1 | grep -v amdgpu.ids | tail -3 | tee $OUT/tune.txt
[ -s $OUT/tunableop_gfx950_lxmert_pert.csv ] && cp $OUT/tunableop_gfx950_lxmert_pert.csv $CSV
python tools/probe_lxmert_pert_trace.py 32 5 2>&1 | grep "image test" | tee $OUT/after.txt
timeout 300 python -m pytest tests/test_gpu_perturbation.py -q -m gpu -x 2>&1 | tail -3 | tee $OUT/pytest.txt
