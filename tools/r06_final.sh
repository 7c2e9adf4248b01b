#!/bin/bash
# what the driver runs at round end: smoke, the whole GPU suite, the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06final; mkdir -p $O
timeout 600 python __graft_entry__.py --smoke > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.txt
timeout 3000 python -m pytest tests -m gpu -x -q --timeout 600 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06final/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic_bytes_per_step'], d['gt_semantics_step']['ms_per_step'], d['update_step']['ms_per_update'])
PY
