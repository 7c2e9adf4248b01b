#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "transpose" --timeout 300 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_rednet.py tests/test_gpu_predsem.py -x -q --timeout 300 2>&1 | tail -5
P5="--no-update --no-collect --no-gt-leg --no-cpu-baseline --reps 3"
for v in "IVLN_X=1" "IVLN_BF3_CONVT=0" "IVLN_X=1"; do
  env $v timeout 300 python bench.py $P5 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$v /"
done > $O/predsem_ab2.txt 2>&1
cat $O/predsem_ab2.txt
IVLN_REDNET_PLAN=0 timeout 200 python tools/gemm_shapes.py rednet 2>&1 | grep " 1  3 \| 8  3 "
