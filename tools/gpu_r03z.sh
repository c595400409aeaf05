#!/bin/bash
# r03z: full gpu suite (schedule-equivalence tests, one-read-per-image mask generator, small_linear), DETR evaluator, bench legs
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03z; mkdir -p $OUT
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 | tee $OUT/pytest.txt
timeout 300 python examples/detr_masks_eval.py --num-images 48 2>&1 | tail -1 | tee $OUT/detr_eval.txt
timeout 300 python examples/detr_masks_eval.py --num-images 48 --graph-slots 0 2>&1 | tail -1 | tee -a $OUT/detr_eval.txt
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --legs cfg3,cfg4 > $OUT/bench.json 2> $OUT/bench.log; tail -2 $OUT/bench.log
python - <<'P'
import json
d=json.loads(open("gpurun_out/r03z/bench.json").read().strip().splitlines()[-1])
for k,v in d["configs"].items(): print(k, {kk:vv for kk,vv in v.items() if kk in ("rate","ms","K10","K20","explain_ms","perturb_ms")})
P
