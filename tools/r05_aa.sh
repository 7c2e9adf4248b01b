#!/bin/bash
# Round 5: decoder skip adds in the 1x1 epilogue (residual behind the ReLU) - tests, A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05aa
rm -rf $O && mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_rednet.py tests/test_gpu_predsem.py -m gpu -q > $O/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -6 $O/pytest_kernels.log | cut -c1-200
P="--no-update --no-collect --no-gt-leg --no-cpu-baseline --reps 3"
for v in "IVLN_REDNET_SKIP_ADD=0" "IVLN_X=1" "IVLN_REDNET_SKIP_ADD=0" "IVLN_X=1"; do
  env $v timeout 300 python bench.py $P 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s|^|$v |"
done > $O/pred_ab.txt 2>&1
cat $O/pred_ab.txt
