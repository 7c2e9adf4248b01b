#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05am
rm -rf $O && mkdir -p $O
for i in 1 2; do timeout 300 python bench.py --only-update --steps 10 2>/dev/null | grep -o '"ms_per_update": [0-9.]*' | head -1; done
timeout 300 python tools/gemm_shapes.py update 2>&1 | grep -v amdgpu.ids | head -16 | cut -c1-110
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train.py tests/test_gpu_policy.py tests/test_gpu_rednet.py tests/test_gpu_predsem.py -m gpu -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -3 $O/pytest.log | cut -c1-200
P="--no-update --no-collect --no-cpu-baseline --reps 3"
timeout 400 python bench.py $P 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -2
