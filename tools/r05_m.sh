#!/bin/bash
# Round 5: 32-pixel tiles for the half-chip 3x3 K-split launches; timeline of the tiled kernel's starved shapes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05m
rm -rf $O && mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_rednet.py tests/test_gpu_predsem.py -m gpu -q > $O/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -4 $O/pytest_kernels.log | cut -c1-200
timeout 600 python tools/conv_bf3_phases.py 2>&1 | grep -v amdgpu.ids | grep -E "rednet|shape" | head -9 | cut -c1-330 > $O/tiled_timeline.txt; cat $O/tiled_timeline.txt
for v in 1 2; do echo "IVLN_BF3_KS_TN=$v"; IVLN_BF3_KS_TN=$v timeout 600 python tools/conv_bf3_ks_phases.py 2>&1 | grep -A8 "3x3 shape" | cut -c1-170; done > $O/ks3_phases.txt; cat $O/ks3_phases.txt
P="--no-update --no-collect --no-gt-leg --no-cpu-baseline --reps 3"
for v in "IVLN_BF3_KS_TN=2" "IVLN_X=1" "IVLN_BF3_KS_TN=2" "IVLN_X=1"; do
  env $v timeout 300 python bench.py $P 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s|^|$v |"
done > $O/pred_ab.txt 2>&1
cat $O/pred_ab.txt
