#!/bin/bash
# Round-3 profile set (run on the GPU box through gpurun; outputs under gpurun_out/r03/, copied to profiles/r03_* by hand).
# Kernel-trace passes and PMC passes are separate runs; every run is bounded by a timeout.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03
rm -rf $O && mkdir -p $O
COMMON="--no-cpu-baseline --no-update --no-pred-leg --no-collect"
run() {  # name, rocprof args..., -- program args
  name=$1; shift
  timeout -k 10 ${T:-170} rocprofv3 "$@" > $O/$name.log 2>&1
  echo "$name rc=$?"
}
stats() { f=$(find $O/$1 -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_stats.py $f $O/$2 > /dev/null; }
pmc() { f=$(find $O/$1 -name "*.db" | head -1); [ -n "$f" ] && python tools/pmc_stats.py $f $O/$2 > /dev/null; }
# --- kernel traces ---
run graph   --kernel-trace -d $O/graph   -- python3 bench.py --steps 200 --warmup 20 $COMMON;            stats graph rollout_graph_kernel_stats.csv
run eager   --kernel-trace -d $O/eager   -- python3 bench.py --steps 200 --warmup 20 $COMMON --no-graph; stats eager rollout_eager_kernel_stats.csv
run predsem --kernel-trace -d $O/predsem -- python3 bench.py --pred-semantics --pred-envs 8 --steps 50 --warmup 5 $COMMON; stats predsem predsem_B8_graph_kernel_stats.csv
run update  --kernel-trace -d $O/update  -- python3 bench.py --only-update --steps 5;                    stats update update_T64N8_kernel_stats.csv
run rednet  --kernel-trace -d $O/rednet  -- python3 tools/bench_components.py rednet;                   stats rednet rednet_B8_kernel_stats.csv
run collect --kernel-trace -d $O/collect -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-update --no-pred-leg; stats collect collect_B8_graph_kernel_stats.csv
grep -h '"metric"' $O/graph.log | tail -1 > $O/bench_graph.json
grep -h '"metric"' $O/predsem.log | tail -1 > $O/bench_predsem_B8.json
# --- PMC passes: one counter per run, kernel trace only ---
for c in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | tr A-Z a-z)
  run r_$n --kernel-trace --pmc $c -d $O/r_$n -- python3 bench.py --steps 20 --warmup 5 $COMMON --no-graph;  pmc r_$n rollout_pmc_$n.csv
  run p_$n --kernel-trace --pmc $c -d $O/p_$n -- python3 bench.py --pred-semantics --pred-envs 8 --steps 10 --warmup 2 $COMMON --no-graph; pmc p_$n predsem_B8_pmc_$n.csv
  run u_$n --kernel-trace --pmc $c -d $O/u_$n -- python3 bench.py --only-update --steps 5;                    pmc u_$n update_pmc_$n.csv
done
run mfma --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $O/u_mfma -- python3 bench.py --only-update --steps 5; pmc u_mfma update_pmc_mfma_util.csv
run pmfma --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $O/p_mfma -- python3 bench.py --pred-semantics --pred-envs 8 --steps 10 --warmup 2 $COMMON --no-graph; pmc p_mfma predsem_B8_pmc_mfma_util.csv
find $O -name "*.db" -delete
# steps traced per rollout pass = warm-up 5 + timed 20 + instrumented roofline passes 20 + 20 (mapper); pred-sem: 2 + 10 + 6
python tools/pmc_traffic.py $O/rollout_pmc_fetch_size.csv $O/rollout_pmc_write_size.csv 65 $O/rollout_pmc_traffic.json "--envs 4 --steps 20 --warmup 5"
python tools/pmc_traffic.py $O/predsem_B8_pmc_fetch_size.csv $O/predsem_B8_pmc_write_size.csv 18 $O/predsem_B8_pmc_traffic.json "--pred-semantics --pred-envs 8 --steps 10 --warmup 2"
# the driver's line, un-profiled
timeout 600 python bench.py > $O/bench_full.json 2> $O/bench_full.err
ls -la $O | grep -v "^d"
