#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03au; mkdir -p $OUT
timeout 85 python bench.py --legs cfg1,cfg4 --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_check.json 2> $OUT/bench.log; tail -1 $OUT/bench.log; cut -c1-300 $OUT/bench_check.json
