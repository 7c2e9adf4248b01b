#!/bin/bash
# round 6, call C: which test of the pred-semantics files hangs; the transposed-conv forms; the recapture test
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06c; mkdir -p $O
timeout 120 tools/store_hazard 64 > $O/store_hazard.txt 2>&1; cat $O/store_hazard.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "transpose or stride2_1x1" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_rednet.py tests/test_gpu_predsem.py -x -v --timeout 150 2>&1 | tail -60 > $O/predsem_tests.txt; tail -40 $O/predsem_tests.txt
timeout 600 python -m pytest tests/test_gpu_policy.py -x -q --timeout 200 2>&1 | tail -15
