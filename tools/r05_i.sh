#!/bin/bash
# Round 5: L2 touches in the 1x1 wave-split kernels; per-wave phases (1x1 and 3x3); stream priorities of the split replay
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05i
rm -rf $O && mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_rednet.py tests/test_gpu_predsem.py -m gpu -q > $O/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -4 $O/pytest_kernels.log | cut -c1-200
for v in 1 0; do IVLN_BF3_TOUCH=$v timeout 600 python tools/conv_bf3_ks_phases.py 2>&1 | grep -v amdgpu.ids > $O/ks_phases_touch$v.txt; cat $O/ks_phases_touch$v.txt; done
P="--no-update --no-collect --no-gt-leg --no-cpu-baseline --reps 3"
for v in "IVLN_BF3_TOUCH=0" "IVLN_X=1" "IVLN_BF3_TOUCH=0" "IVLN_X=1" "IVLN_DEPTH_STREAM_PRIORITY=0" "IVLN_DEPTH_STREAM_PRIORITY=0 IVLN_MAIN_STREAM_PRIORITY=-1" "IVLN_MAIN_STREAM_PRIORITY=0" "IVLN_X=1"; do
  env $v timeout 300 python bench.py $P 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s|^|$v |"
done > $O/pred_ab.txt 2>&1
cat $O/pred_ab.txt
timeout 300 python tools/gemm_shapes.py rednet > $O/rednet_gemm_shapes.txt 2>&1; head -48 $O/rednet_gemm_shapes.txt | cut -c1-110
