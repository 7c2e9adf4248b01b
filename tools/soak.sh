#!/bin/bash
# fresh-process soak of the split-graph replay: N short bench runs, report any that exceed the time limit
n=${1:-20}; bad=0
for i in $(seq 1 $n); do
  s=$(date +%s)
  timeout -k 5 100 python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-update --no-pred-leg > gpurun_out/soak_$i.log 2>&1
  rc=$?; e=$(date +%s)
  v=$(tail -1 gpurun_out/soak_$i.log | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])" 2>/dev/null)
  echo "run $i rc=$rc $((e-s))s value=$v"
  [ $rc -ne 0 ] && bad=$((bad+1))
done
echo "bad=$bad"
