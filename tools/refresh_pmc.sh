#!/bin/bash
# PMC passes (separate runs per counter): rollout FETCH/WRITE (eager), update FETCH/WRITE
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/pmc; rm -rf $O; mkdir -p $O
COMMON="--no-cpu-baseline --no-update --no-pred-leg --no-graph"
for c in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | tr A-Z a-z)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c -d $O/r_$n -- python3 bench.py --steps 20 --warmup 5 $COMMON > $O/r_$n.log 2>&1; echo "rollout $c rc=$?"
  f=$(find $O/r_$n -name "*.db" | head -1); python tools/pmc_stats.py $f $O/rollout_pmc_$n.csv > /dev/null
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c -d $O/u_$n -- python3 bench.py --only-update --steps 5 > $O/u_$n.log 2>&1; echo "update $c rc=$?"
  f=$(find $O/u_$n -name "*.db" | head -1); python tools/pmc_stats.py $f $O/update_pmc_$n.csv > /dev/null
done
find $O -name "*.db" -delete
# steps traced per pass = warm-up 5 + timed 20 + instrumented roofline passes 20 + 20 (mapper)
python tools/pmc_traffic.py $O/rollout_pmc_fetch_size.csv $O/rollout_pmc_write_size.csv ${PMC_STEPS:-65} $O/rollout_pmc_traffic.json
