#!/bin/bash
# Round 5: fused bottleneck tail with guarded stores - A/B, then the whole GPU suite
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05u
rm -rf $O && mkdir -p $O
P="--no-update --no-collect --no-gt-leg --no-cpu-baseline --reps 3"
for v in "IVLN_BF3_FUSE=0" "IVLN_X=1" "IVLN_BF3_FUSE=0" "IVLN_X=1"; do
  env $v timeout 300 python bench.py $P 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s|^|$v |"
done > $O/pred_ab.txt 2>&1
cat $O/pred_ab.txt
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -5 $O/pytest_gpu.log | cut -c1-200
