#!/bin/bash
# pred-semantics step (RedNet at 4 envs) against the non-deferred split-K policy knobs
for cfg in "512 512" "512 1024" "256 1024" "1024 1024" "256 768"; do
  set -- $cfg
  r=$(IVLN_SPLIT_BELOW=$1 IVLN_SPLIT_WANT_ND=$2 timeout -k 5 120 python bench.py --pred-semantics --steps 100 --warmup 10 --no-update --no-cpu-baseline --no-pred-leg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "below=$1 want=$2 -> $r"
done
