#!/bin/bash
# Round 5, second GPU call: the GPU test suite, conv_bf3 tile sweep on the new staging, torch ops left in the update, bench.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05b
rm -rf $O && mkdir -p $O
timeout 300 python tools/conv_bf3_probe.py all > $O/conv_bf3_probe.txt 2>&1
for c in 2 3 4 5 6; do IVLN_SPLIT_BF16_CFG=$c timeout 300 python tools/conv_bf3_probe.py rednet > $O/conv_bf3_probe_cfg$c.txt 2>&1; done
timeout 300 python tools/update_torch_ops.py > $O/update_torch_ops.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
timeout 700 python bench.py --no-cpu-baseline > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$?"
IVLN_NO_WGRAD_OVERLAP=1 timeout 300 python bench.py --only-update --steps 10 > $O/bench_update_nooverlap.json 2>&1
timeout 300 python bench.py --only-update --steps 10 > $O/bench_update.json 2>&1
ls -la $O
