#!/bin/bash
# stability soak: the graph-heavy GPU test files several times in fresh processes, then the whole suite once more; a failing
# iteration leaves its full output under gpurun_out/soak/ and its failure lines on stdout
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/soak; mkdir -p $O
N=${1:-6}
for i in $(seq 1 $N); do
  timeout 900 python -m pytest tests/test_gpu_policy.py tests/test_gpu_predsem.py tests/test_gpu_train.py tests/test_gpu_depth_net.py -x -q --timeout 300 > $O/iter_$i.txt 2>&1
  rc=$?; echo "iteration $i rc=$rc: $(tail -1 $O/iter_$i.txt)"
  if [ $rc -ne 0 ]; then grep -nE "^E |Error|FAILED|assert" $O/iter_$i.txt | head -40; else rm -f $O/iter_$i.txt; fi
done
if [ "${2:-suite}" = suite ]; then timeout 3000 python -m pytest tests -m gpu -x -q --timeout 600 2>&1 | tail -2; fi
