#!/bin/bash
# round 6, call H: mapper prefix beside RedNet (probe + A/B), twin / mapper / graph tests, the 8-rank one-device bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_twin.py tests/test_gpu_mapper.py tests/test_gpu_predsem.py tests/test_gpu_policy.py -x -q --timeout 200 2>&1 | tail -8
for v in "IVLN_X=1" "IVLN_MAPPER_PREFIX=0"; do echo "== $v"; env $v timeout 200 python tools/split_probe.py pred 8 2>&1 | tail -2; done > $O/split_probe_prefix.txt 2>&1; cat $O/split_probe_prefix.txt
P5="--no-update --no-collect --no-gt-leg --no-cpu-baseline --reps 3"
for v in "IVLN_X=1" "IVLN_MAPPER_PREFIX=0" "IVLN_X=1" "IVLN_MAPPER_PREFIX=0"; do
  env $v timeout 300 python bench.py $P5 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$v /"
done > $O/predsem_ab.txt 2>&1
cat $O/predsem_ab.txt
timeout 900 python -m pytest tests/test_gpu_bench.py -x -q --timeout 900 2>&1 | tail -8
