#!/bin/bash
# Round 5: fused bottleneck tail (conv2 3x3 + conv3 1x1 in one launch) + register-direct 1x1 epilogue: tests, A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05s
rm -rf $O && mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_rednet.py tests/test_gpu_predsem.py -m gpu -q > $O/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -6 $O/pytest_kernels.log | cut -c1-200
P="--no-update --no-collect --no-gt-leg --no-cpu-baseline --reps 3"
for v in "IVLN_BF3_FUSE=0" "IVLN_X=1" "IVLN_BF3_FUSE=0" "IVLN_X=1"; do
  env $v timeout 300 python bench.py $P 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s|^|$v |"
done > $O/pred_ab.txt 2>&1
cat $O/pred_ab.txt
