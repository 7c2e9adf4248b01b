#!/bin/bash
# Round 5: fused bottleneck tail with the 128 x 64-pixel tile for layer 2 - tests, race check, A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05x
rm -rf $O && mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_rednet.py tests/test_gpu_predsem.py -m gpu -q > $O/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -6 $O/pytest_kernels.log | cut -c1-200
for i in 1 2; do timeout 300 python tools/dbg_fuse.py 2>&1 | grep -E "BAD|scores" | cut -c1-300; done
P="--no-update --no-collect --no-gt-leg --no-cpu-baseline --reps 3"
for v in "IVLN_BF3_FUSE_PX=128" "IVLN_X=1" "IVLN_BF3_FUSE=0" "IVLN_BF3_FUSE_PX=128" "IVLN_X=1"; do
  env $v timeout 300 python bench.py $P 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s|^|$v |"
done > $O/pred_ab.txt 2>&1
cat $O/pred_ab.txt
