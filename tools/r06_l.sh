#!/bin/bash
# round 6, call L: stride-2 3x3 convs on the split-bf16 kernel (phase planes)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "stride2_3x3 or test_conv2d_split_bf16_kernel" --timeout 300 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_rednet.py tests/test_gpu_predsem.py -x -q --timeout 300 2>&1 | tail -5
P5="--no-update --no-collect --no-gt-leg --no-cpu-baseline --reps 3"
for v in "IVLN_X=1" "IVLN_BF3_S2=0" "IVLN_X=1" "IVLN_BF3_S2=0"; do
  env $v timeout 300 python bench.py $P5 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$v /"
done > $O/predsem_ab.txt 2>&1
cat $O/predsem_ab.txt
IVLN_REDNET_PLAN=0 timeout 200 python tools/gemm_shapes.py rednet 2>&1 | grep " 2    9 \|  7  0  0 2" 
