#!/bin/bash
# stability soak: the graph-heavy GPU test files several times in fresh processes, then the whole suite once more
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/soak; mkdir -p $O
for i in 1 2 3 4 5 6; do
  timeout 900 python -m pytest tests/test_gpu_policy.py tests/test_gpu_predsem.py tests/test_gpu_train.py tests/test_gpu_depth_net.py -x -q --timeout 300 2>&1 | tail -1
done
timeout 3000 python -m pytest tests -m gpu -x -q --timeout 600 2>&1 | tail -2
