#!/bin/bash
# round 6, call F: PMC passes of the pred-semantics step (r06 traffic json), then the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06f; mkdir -p $O
bash tools/refresh_r06.sh pmc > $O/pmc.log 2>&1; tail -3 $O/pmc.log
timeout 2700 python -m pytest tests -m gpu -x -q --timeout 600 > $O/pytest_gpu.txt 2>&1; tail -15 $O/pytest_gpu.txt
