#!/bin/bash
# r03p: full gpu suite after (a) top-block forward on the output rows, (b) sweep gelu-bwd / row-resident LN-bwd bf16 kernels,
# (c) LXMERT / VisualBERT relprop; bench with config legs; LRP methods through the evaluators
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03p; mkdir -p $OUT
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 | tee $OUT/pytest.txt
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.log; tail -3 $OUT/bench.log; cut -c1-300 $OUT/bench.json
timeout 300 python examples/lxmert_perturbation_eval.py --method ours_with_lrp --num-samples 8 2>&1 | tail -2 | tee $OUT/lxmert_lrp_eval.txt
timeout 300 python examples/lxmert_perturbation_eval.py --method transformer_att --num-samples 8 --is-text-pert True 2>&1 | tail -2 | tee -a $OUT/lxmert_lrp_eval.txt
timeout 300 python examples/visualbert_pert_eval.py --method transformer_attribution --num-samples 8 2>&1 | tail -2 | tee $OUT/visualbert_lrp_eval.txt
timeout 300 python examples/visualbert_pert_eval.py --method partial_lrp --num-samples 8 2>&1 | tail -2 | tee -a $OUT/visualbert_lrp_eval.txt
