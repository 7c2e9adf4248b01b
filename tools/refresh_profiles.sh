#!/bin/bash
# Re-collect the rocprofv3 summaries committed under profiles/ (run on the GPU box through gpurun; outputs under
# gpurun_out/refresh/).  Kernel-trace passes and PMC passes are separate runs; every run is bounded by a timeout.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/refresh
rm -rf $O && mkdir -p $O
COMMON="--no-cpu-baseline --no-update --no-pred-leg"
run() {  # name, rocprof args..., -- program args
  name=$1; shift
  timeout -k 10 150 rocprofv3 "$@" > $O/$name.log 2>&1
  echo "$name rc=$?"
}
run graph   --kernel-trace -d $O/graph   -- python3 bench.py --steps 200 --warmup 20 $COMMON
run eager   --kernel-trace -d $O/eager   -- python3 bench.py --steps 200 --warmup 20 $COMMON --no-graph
run predsem --kernel-trace -d $O/predsem -- python3 bench.py --pred-semantics --steps 50 --warmup 5 $COMMON
run update  --kernel-trace -d $O/update  -- python3 tools/bench_components.py update
run rednet  --kernel-trace -d $O/rednet  -- python3 tools/bench_components.py rednet
run fetch   --kernel-trace --pmc FETCH_SIZE -d $O/fetch -- python3 bench.py --steps 20 --warmup 5 $COMMON --no-graph
run write   --kernel-trace --pmc WRITE_SIZE -d $O/write -- python3 bench.py --steps 20 --warmup 5 $COMMON --no-graph
for n in graph eager predsem update rednet; do
  f=$(find $O/$n -name "*.db" | head -1)
  [ -n "$f" ] && python tools/rocpd_stats.py $f $O/${n}_kernel_stats.csv > /dev/null
done
for n in fetch write; do
  f=$(find $O/$n -name "*.db" | head -1)
  [ -n "$f" ] && python tools/pmc_stats.py $f $O/pmc_${n}_size.csv > /dev/null
done
grep -h '"metric"' $O/graph.log | tail -1 > $O/bench_graph.json
find $O -name "*.db" -delete
ls -la $O
