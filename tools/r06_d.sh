#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
echo "== policy file, capture-stream warm-up ON"
timeout 600 python -m pytest tests/test_gpu_policy.py -x -v --timeout 60 2>&1 | grep -a "PASSED\|FAILED\|Timeout\|passed\|failed\|graphed.py\|ops.py\|depth_net.py" | head -40
echo "== policy file, capture-stream warm-up OFF"
IVLN_CAPTURE_STREAM_WARMUP=0 timeout 600 python -m pytest tests/test_gpu_policy.py -x -v --timeout 60 2>&1 | grep -a "PASSED\|FAILED\|Timeout\|passed\|failed" | head -40
