#!/bin/bash
# A/B of the XCD-aware workgroup remap: update leg, RedNet, rollout
for v in "" 1; do
  echo "== IVLN_NO_XCD_REMAP='$v'"
  IVLN_NO_XCD_REMAP=$v timeout -k 5 200 python tools/bench_components.py update 2>&1 | grep "update T="
  IVLN_NO_XCD_REMAP=$v timeout -k 5 200 python tools/bench_components.py rednet 2>&1 | grep "RedNet fwd"
  IVLN_NO_XCD_REMAP=$v timeout -k 5 200 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-update 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rollout', d['value'], 'pred', d['pred_semantics_step']['value'])"
done
