#!/bin/bash
# Round 5: double-buffered staging A/B in one call + kernel tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05g
rm -rf $O && mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_rednet.py tests/test_gpu_predsem.py -m gpu -q > $O/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -4 $O/pytest_kernels.log | cut -c1-200
for v in 0 1; do IVLN_BF3_DBUF=$v timeout 300 python tools/conv_bf3_probe.py rednet 2>&1 | grep rednet | cut -c1-80 > $O/probe_dbuf$v.txt; done
paste $O/probe_dbuf0.txt $O/probe_dbuf1.txt | cut -c1-75,120-160
P="--no-update --no-collect --no-gt-leg --no-cpu-baseline --reps 3"
for v in "IVLN_BF3_DBUF=0" "IVLN_X=1" "IVLN_BF3_DBUF=0" "IVLN_X=1"; do
  env $v timeout 300 python bench.py $P 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$v /"
done > $O/pred_ab.txt 2>&1
cat $O/pred_ab.txt
for v in "IVLN_BF3_DBUF=0" "IVLN_X=1"; do env $v timeout 300 python bench.py --only-update --steps 10 2>/dev/null | grep -o '"ms_per_update": [0-9.]*' | head -1 | sed "s/^/$v /"; done
