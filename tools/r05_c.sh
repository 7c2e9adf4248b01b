#!/bin/bash
# Round 5, third GPU call: tests; K-split kernel vs tiled (probe); depth encoder beside RedNet three ways; staging A/B on the update
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05c
rm -rf $O && mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log | cut -c1-220
for k in 0 1; do IVLN_BF3_KS=$k timeout 300 python tools/conv_bf3_probe.py rednet > $O/conv_bf3_probe_ks$k.txt 2>&1; done
P="--no-update --no-collect --no-gt-leg --no-cpu-baseline --reps 3"
for m in pairs chain net; do IVLN_PRED_DEPTH=$m timeout 300 python bench.py $P > $O/bench_pred_$m.json 2> $O/bench_pred_$m.err; grep -o '"ms_per_step": [0-9.]*' $O/bench_pred_$m.json | head -1; done
IVLN_BF3_KS=0 timeout 300 python bench.py $P > $O/bench_pred_ks0.json 2> $O/bench_pred_ks0.err; grep -o '"ms_per_step": [0-9.]*' $O/bench_pred_ks0.json | head -1
IVLN_REDNET_PLAN=0 timeout 200 python tools/gemm_shapes.py rednet > $O/rednet_B8_gemm_shapes.txt 2>&1
for i in 1 2 3; do
  IVLN_HIP_LIB=$PWD/tools/ab/libivln_hip_oldstage.so timeout 300 python bench.py --only-update --steps 10 2>/dev/null | grep -o '"ms_per_update": [0-9.]*' | head -1 | sed 's/^/old /'
  timeout 300 python bench.py --only-update --steps 10 2>/dev/null | grep -o '"ms_per_update": [0-9.]*' | head -1 | sed 's/^/new /'
done > $O/update_ab.txt 2>&1
cat $O/update_ab.txt
timeout 700 python bench.py --no-cpu-baseline > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$?"
ls $O
