#!/bin/bash
# Round 5: RedNet side branches forked inside the captured step - parity tests + A/B per branch kind
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05q
rm -rf $O && mkdir -p $O
IVLN_REDNET_FORK=7 timeout 900 python -m pytest tests/test_gpu_rednet.py tests/test_gpu_predsem.py -m gpu -q > $O/pytest_fork.log 2>&1; echo "fork tests rc=$?"; tail -4 $O/pytest_fork.log | cut -c1-200
P="--no-update --no-collect --no-gt-leg --no-cpu-baseline --reps 3"
for v in 0 7 1 2 4 0 7; do
  IVLN_REDNET_FORK=$v timeout 300 python bench.py $P 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s|^|IVLN_REDNET_FORK=$v |"
done > $O/pred_ab.txt 2>&1
cat $O/pred_ab.txt
