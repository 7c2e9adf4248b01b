#!/bin/bash
# Round 5, sixth GPU call: 1x1 tests incl. Cin = 64, launch tally per form, phases of the tiled kernel, A/B in one call
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05f
rm -rf $O && mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "split_bf16 or conv1x1_split" > $O/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -4 $O/pytest_kernels.log | cut -c1-200
timeout 300 python tools/conv_bf3_phases.py > $O/conv_bf3_phases.txt 2>&1; tail -22 $O/conv_bf3_phases.txt
for v in "IVLN_BF3_1X1_KS=0" "IVLN_X=1"; do env $v IVLN_REDNET_PLAN=0 timeout 200 python tools/gemm_shapes.py rednet > $O/shapes_$v.txt 2>&1; done
P="--no-update --no-collect --no-gt-leg --no-cpu-baseline --reps 3"
for v in "IVLN_BF3_1X1_KS=0" "IVLN_X=1" "IVLN_BF3_1X1_FORM=ks" "IVLN_X=1"; do
  env $v timeout 300 python bench.py $P 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$v /"
done > $O/pred_ab.txt 2>&1
cat $O/pred_ab.txt
ls $O
