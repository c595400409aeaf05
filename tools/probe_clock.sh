#!/bin/bash
# What clock does the GPU run at while a tiny latency-bound kernel is replayed?  (rocm-smi sampled beside the schedule probe)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python tools/probe_schedule.py > gpurun_out/clock_probe_sched.txt 2>&1 &
PID=$!
for i in $(seq 1 40); do
  /opt/rocm/bin/rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | head -3 | tr '\n' ' '; echo
  sleep 0.5
  kill -0 $PID 2>/dev/null || break
done
wait $PID
grep -v amdgpu.ids gpurun_out/clock_probe_sched.txt | head -20
