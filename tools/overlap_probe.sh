#!/bin/bash
for v in "" 1; do
  echo "== IVLN_NO_TRAIN_OVERLAP='$v'"
  IVLN_NO_TRAIN_OVERLAP=$v timeout -k 5 200 python tools/bench_components.py update 2>&1 | grep "update T="
  IVLN_NO_TRAIN_OVERLAP=$v timeout -k 5 200 python tools/bench_components.py update 2>&1 | grep "update T="
done
timeout -k 5 300 python -m pytest tests/test_gpu_train.py -x -q -m gpu --timeout 200 2>&1 | tail -2
