#!/bin/bash
# Side measurements of a round (after tools/gpu_round.sh): cfg-5 probes, attention-mode probe, CPU-thread sweep, DETR / LXMERT probes.
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
{ timeout 300 python tools/probe_cfg5.py 2>&1 | grep -v amdgpu.ids; timeout 300 python tools/probe_cfg5.py bf16 gemm body 2>&1 | grep -v amdgpu.ids | tail -4; } > $OUT/cfg5_probe.txt
timeout 200 python tools/probe_attn_bf16.py 64 2>&1 | grep -v amdgpu.ids > $OUT/attn_bf16_probe.txt
for t in 8 16 32 64 128; do timeout 120 python tools/probe_cpu_baseline.py $t 8 2>&1 | grep threads=; done > $OUT/cpu_threads.txt
timeout 300 python tools/probe_detr.py 2>&1 | grep -v amdgpu.ids | tail -30 > $OUT/detr_probe.txt
timeout 300 python tools/probe_lxmert_pert.py 2>&1 | grep -v amdgpu.ids | tail -30 > $OUT/lxmert_probe.txt
tail -3 $OUT/cfg5_probe.txt; cat $OUT/cpu_threads.txt | tail -6
