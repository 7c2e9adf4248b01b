#!/bin/bash
# Round 5, fourth GPU call: new kernels' tests first, deep-K 1x1 probe, pred-semantics bench A/B, then the whole suite
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05d
rm -rf $O && mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "split_bf16 or conv1x1_split" > $O/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -5 $O/pytest_kernels.log | cut -c1-200
for k in 0 1; do IVLN_BF3_1X1_KS=$k timeout 300 python tools/conv_bf3_probe.py one > $O/conv_bf3_probe_1x1ks$k.txt 2>&1; done
timeout 300 python tools/conv_bf3_probe.py rednet > $O/conv_bf3_probe.txt 2>&1
P="--no-update --no-collect --no-gt-leg --no-cpu-baseline --reps 3"
timeout 300 python bench.py $P > $O/bench_pred.json 2> $O/bench_pred.err; grep -o '"ms_per_step": [0-9.]*' $O/bench_pred.json | head -1
IVLN_BF3_1X1_KS=0 timeout 300 python bench.py $P > $O/bench_pred_1x1ks0.json 2> $O/bench_pred_1x1ks0.err; grep -o '"ms_per_step": [0-9.]*' $O/bench_pred_1x1ks0.json | head -1
IVLN_REDNET_PLAN=0 timeout 200 python tools/gemm_shapes.py rednet > $O/rednet_B8_gemm_shapes.txt 2>&1
timeout 3000 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log | cut -c1-220
ls $O
