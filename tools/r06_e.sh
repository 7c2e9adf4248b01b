#!/bin/bash
# round 6, call E: the whole GPU suite (twice for the graph tests: an intermittent capture problem would show), transposed-conv A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06e; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 300 > $O/pytest_gpu.txt 2>&1; tail -8 $O/pytest_gpu.txt
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_policy.py tests/test_gpu_predsem.py tests/test_gpu_rednet.py -x -q --timeout 120 2>&1 | tail -2; done
P5="--no-update --no-collect --no-gt-leg --no-cpu-baseline --reps 3"
for v in "IVLN_X=1" "IVLN_BF3_CONVT=0" "IVLN_PRED_DEPTH_START=layer4" "IVLN_PRED_DEPTH_START=layer4 IVLN_BF3_CONVT=0" "IVLN_PRED_DEPTH_START=layer3" "IVLN_X=1"; do
  env $v timeout 300 python bench.py $P5 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$v /"
done > $O/predsem_ab.txt 2>&1
cat $O/predsem_ab.txt
