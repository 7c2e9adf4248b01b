#!/bin/bash
# fresh-process soak of the split-graph replay (persistent depth encoder beside the mapper / map-CNN graph, over-subscribed bi-LSTM
# grid): N short bench runs of the gt-semantics step at 4 and 8 envs + the collection / update alternation, report any that fail
# or exceed the time limit
n=${1:-10}; bad=0
mkdir -p gpurun_out
for i in $(seq 1 $n); do
  for envs in 4 8; do
    s=$(date +%s)
    timeout -k 5 120 python bench.py --gt-semantics --envs $envs --steps 1500 --warmup 20 --reps 2 --no-cpu-baseline --no-update --no-collect --no-pred-leg > gpurun_out/soak_${i}_$envs.log 2>&1
    rc=$?; e=$(date +%s)
    v=$(tail -1 gpurun_out/soak_${i}_$envs.log | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
    echo "run $i envs=$envs rc=$rc $((e-s))s ms_per_step=$v"
    [ $rc -ne 0 ] && bad=$((bad+1))
  done
done
s=$(date +%s)
timeout -k 5 300 python bench.py --no-cpu-baseline --reps 3 > gpurun_out/soak_full.log 2>&1
echo "full bench rc=$? $(( $(date +%s) - s ))s"
echo "bad=$bad"
