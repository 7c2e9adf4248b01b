#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06g; mkdir -p $O
IVLN_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 8 --steps 3 --warmup 1 --reps 1 --gt-semantics --no-pred-leg --no-collect --no-cpu-baseline > $O/b8.out 2> $O/b8.err; echo "rc=$?"
grep -v "^\[bench\]\|Gloo\|amdgpu.ids" $O/b8.err | head -60
tail -c 600 $O/b8.out
timeout 2700 python -m pytest tests -m gpu -x -q --timeout 600 --deselect tests/test_gpu_bench.py::test_bench_gpus_8_on_one_device_runs_the_update_collective_over_8_ranks > $O/pytest_gpu.txt 2>&1; tail -15 $O/pytest_gpu.txt
