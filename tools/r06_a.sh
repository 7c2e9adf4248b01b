#!/bin/bash
# round 6, call A: where the depth encoder's graph should start beside RedNet (pred-semantics step, 8 envs), one box
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06a; mkdir -p $O
for st in 0 stem layer1 layer2 layer3 layer4 deconv1 deconv2; do
  for pr in -1 0; do
    echo "== IVLN_PRED_DEPTH_START=$st IVLN_PRED_DEPTH_PRIORITY=$pr"
    IVLN_PRED_DEPTH_START=$st IVLN_PRED_DEPTH_PRIORITY=$pr timeout 200 python tools/split_probe.py pred 8 2>&1 | tail -2
  done
done > $O/split_probe_start.txt 2>&1
cat $O/split_probe_start.txt
P5="--no-update --no-collect --no-gt-leg --no-cpu-baseline --reps 3"
for st in 0 layer2 layer3 layer4; do
  IVLN_PRED_DEPTH_START=$st timeout 300 python bench.py $P5 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/start=$st /"
done > $O/bench_start.txt 2>&1
cat $O/bench_start.txt
timeout 900 python -m pytest tests/test_gpu_predsem.py tests/test_gpu_policy.py -x -q 2>&1 | tail -5
