#!/bin/bash
# One GPU-box round: gpu tests, smoke, bench (JSON line), rocprofv3 kernel trace + PMC pass. Outputs under gpurun_out/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r01
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r01/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/r01/smoke.txt
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r01/bench.json 2> gpurun_out/r01/bench.log; tail -3 gpurun_out/r01/bench.log; cut -c1-600 gpurun_out/r01/bench.json
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r01/trace -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r01/bench_traced.json 2> gpurun_out/r01/trace.log
python tools/prof_summary.py gpurun_out/r01/trace/bench_results.db > gpurun_out/r01/kernel_stats.txt 2>&1; head -30 gpurun_out/r01/kernel_stats.txt
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r01/pmc_fetch -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r01/pmc_fetch.log
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r01/pmc_write -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r01/pmc_write.log
ls -la gpurun_out/r01/pmc_fetch gpurun_out/r01/pmc_write 2>&1 | head; 
