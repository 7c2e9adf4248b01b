#!/bin/bash
# round 6, call B: the store-hazard reproduction; stride-2 1x1 gather A/B; the depth encoder's form beside RedNet with a late start
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06b; mkdir -p $O
timeout 300 tools/store_hazard 256 > $O/store_hazard.txt 2>&1; cat $O/store_hazard.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "stride2_1x1 or conv2d or pools" 2>&1 | tail -3
P5="--no-update --no-collect --no-gt-leg --no-cpu-baseline --reps 3"
for v in "IVLN_X=1" "IVLN_S2_GATHER=0" "IVLN_PRED_DEPTH_START=layer4" "IVLN_PRED_DEPTH_START=layer4 IVLN_S2_GATHER=0" \
         "IVLN_PRED_DEPTH_START=layer4 IVLN_PRED_DEPTH=chain" "IVLN_PRED_DEPTH_START=layer3 IVLN_PRED_DEPTH=chain" \
         "IVLN_PRED_DEPTH_START=layer4 IVLN_PRED_DEPTH=net" "IVLN_PRED_DEPTH_START=deconv1 IVLN_PRED_DEPTH=net" "IVLN_X=1"; do
  env $v timeout 300 python bench.py $P5 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$v /"
done > $O/predsem_ab.txt 2>&1
cat $O/predsem_ab.txt
timeout 900 python -m pytest tests/test_gpu_rednet.py tests/test_gpu_predsem.py -x -q 2>&1 | tail -3
