#!/bin/bash
# Round 5, first GPU call: the GPU test suite, the driver's bench line, and baseline probes for the RedNet step.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05a
rm -rf $O && mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 700 python bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$?"
timeout 200 python tools/split_probe.py pred 8 > $O/split_probe_pred8.txt 2>&1
timeout 200 python tools/split_probe.py gt 4 > $O/split_probe_gt4.txt 2>&1
timeout 200 python tools/split_probe.py gt 8 > $O/split_probe_gt8.txt 2>&1
IVLN_CACHE_INSTRUCTION=0 timeout 200 python tools/split_probe.py gt 4 > $O/split_probe_gt4_nocache.txt 2>&1
IVLN_CACHE_INSTRUCTION=0 timeout 200 python tools/split_probe.py gt 8 > $O/split_probe_gt8_nocache.txt 2>&1
IVLN_REDNET_PLAN=0 timeout 200 python tools/gemm_shapes.py rednet > $O/rednet_B8_gemm_shapes.txt 2>&1
timeout 300 python tools/conv_bf3_probe.py rednet > $O/conv_bf3_probe.txt 2>&1
for c in 1 2 4; do IVLN_SPLIT_BF16_CFG=$c timeout 300 python tools/conv_bf3_probe.py rednet > $O/conv_bf3_probe_cfg$c.txt 2>&1; done
ls -la $O
