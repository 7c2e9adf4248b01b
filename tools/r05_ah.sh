#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05ah
rm -rf $O && mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_rednet.py tests/test_gpu_predsem.py tests/test_gpu_gn_conv.py -m gpu -q > $O/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -3 $O/pytest_kernels.log | cut -c1-200
P="--no-update --no-collect --no-gt-leg --no-cpu-baseline --reps 3"
OLD="IVLN_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/lib_r05_pre_wt32.so"
for v in "$OLD" "IVLN_X=1" "$OLD" "IVLN_X=1"; do
  env $v timeout 300 python bench.py $P 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s|^|${v##*/} |"
done > $O/pred_ab.txt 2>&1
cat $O/pred_ab.txt
