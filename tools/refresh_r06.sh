#!/bin/bash
# Round-6 profile set (run on the GPU box through gpurun; outputs under gpurun_out/r06/, the summaries are ALSO written into
# profiles/r06_* of the box's copy so that tests run in the same call see them; copy gpurun_out/r06/* to profiles/ by hand).
#   tools/refresh_r06.sh pmc     PMC passes of the pred-semantics step only (traffic json + MFMA-busy table)
#   tools/refresh_r06.sh full    everything (kernel traces, PMC passes of all three legs, bench line, probes)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
MODE=${1:-full}
O=gpurun_out/r06
mkdir -p $O
ONLY="--no-cpu-baseline --no-update --no-collect"
PRED="$ONLY --no-gt-leg"
GT="$ONLY --gt-semantics --no-pred-leg"
run() {  # name, rocprof args..., -- program args
  name=$1; shift
  timeout -k 10 ${T:-200} rocprofv3 "$@" > $O/$name.log 2>&1
  echo "$name rc=$?"
}
stats() { f=$(find $O/$1 -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_stats.py $f $O/$2 > /dev/null; }
pmc() { f=$(find $O/$1 -name "*.db" | head -1); [ -n "$f" ] && python tools/pmc_stats.py $f $O/$2 > /dev/null; }
keep() { for f in "$@"; do [ -s $O/$f ] && cp $O/$f profiles/r06_$f; done; }

pred_pmc() {
  for c in FETCH_SIZE WRITE_SIZE; do
    n=$(echo $c | tr A-Z a-z)
    run p_$n --kernel-trace --pmc $c -d $O/p_$n -- python3 bench.py --steps 10 --warmup 2 --reps 1 $PRED --no-graph;  pmc p_$n predsem_B8_pmc_$n.csv
  done
  run pmfma --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $O/p_mfma -- python3 bench.py --steps 10 --warmup 2 --reps 1 $PRED --no-graph; pmc p_mfma predsem_B8_pmc_mfma_util.csv
  # steps traced per pass: pred = warm-up 2 + timed 10 + instrumented roofline pass 6 = 18
  python tools/pmc_traffic.py $O/predsem_B8_pmc_fetch_size.csv $O/predsem_B8_pmc_write_size.csv 18 $O/predsem_B8_pmc_traffic.json "--pred-envs 8 --steps 10 --warmup 2 --reps 1"
  keep predsem_B8_pmc_fetch_size.csv predsem_B8_pmc_write_size.csv predsem_B8_pmc_mfma_util.csv predsem_B8_pmc_traffic.json
}

if [ "$MODE" = pmc ]; then
  pred_pmc
  find $O -name "*.db" -delete
  exit 0
fi

# --- un-profiled runs first (a clean GPU): the driver's line, probes ---
timeout 700 python bench.py > $O/bench_full.json 2> $O/bench_full.err
timeout 300 python bench.py --gt-semantics --envs 8 --no-update --no-collect --no-pred-leg --no-cpu-baseline > $O/bench_gt_B8.json 2> $O/bench_gt_B8.err
timeout 200 python tools/depth_net_phases.py 4 > $O/depth_net_phases_N4.txt 2>&1
for B in 4 8; do echo "== gt envs $B"; timeout 200 python tools/split_probe.py gt $B 2>&1 | tail -2; done > $O/split_probe.txt 2>&1
echo "== pred envs 8" >> $O/split_probe.txt; timeout 200 python tools/split_probe.py pred 8 2>&1 | tail -2 >> $O/split_probe.txt
IVLN_REDNET_PLAN=0 timeout 200 python tools/gemm_shapes.py rednet > $O/rednet_B8_gemm_shapes.txt 2>&1
timeout 300 python tools/update_torch_ops.py > $O/update_torch_ops.txt 2>&1
# --- kernel traces ---
run predsem --kernel-trace -d $O/predsem -- python3 bench.py --steps 50 --warmup 5 --reps 1 $PRED;               stats predsem predsem_B8_graph_kernel_stats.csv
f=$(find $O/predsem -name "*.db" | head -1); [ -n "$f" ] && python tools/step_timeline.py $f > $O/predsem_B8_step_timeline.txt 2>&1
run graph   --kernel-trace -d $O/graph   -- python3 bench.py --steps 200 --warmup 20 --reps 1 $GT;               stats graph rollout_graph_kernel_stats.csv
run update  --kernel-trace -d $O/update  -- python3 bench.py --only-update --steps 5;                            stats update update_T64N8_kernel_stats.csv
grep -h '"metric"' $O/predsem.log | tail -1 > $O/bench_predsem_B8.json
grep -h '"metric"' $O/graph.log | tail -1 > $O/bench_gt_graph.json
# --- PMC passes: one counter per run, kernel trace only ---
pred_pmc
for c in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | tr A-Z a-z)
  run r_$n --kernel-trace --pmc $c -d $O/r_$n -- python3 bench.py --steps 20 --warmup 5 --reps 1 $GT --no-graph;    pmc r_$n rollout_pmc_$n.csv
  run u_$n --kernel-trace --pmc $c -d $O/u_$n -- python3 bench.py --only-update --steps 5;                          pmc u_$n update_pmc_$n.csv
done
run rmfma --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $O/r_mfma -- python3 bench.py --steps 20 --warmup 5 --reps 1 $GT --no-graph;  pmc r_mfma rollout_pmc_mfma_util.csv
run umfma --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $O/u_mfma -- python3 bench.py --only-update --steps 5;                         pmc u_mfma update_pmc_mfma_util.csv
find $O -name "*.db" -delete
# steps traced per pass: gt = warm-up 5 + timed 20 + instrumented roofline passes 20 + 20 (mapper) = 65; update = 2 + 5 + 1 = 8
python tools/pmc_traffic.py $O/rollout_pmc_fetch_size.csv $O/rollout_pmc_write_size.csv 65 $O/rollout_pmc_traffic.json "--gt-semantics --envs 4 --steps 20 --warmup 5 --reps 1 (depth encoder = the persistent launch, eager)" 45
python tools/pmc_traffic.py $O/update_pmc_fetch_size.csv $O/update_pmc_write_size.csv 8 $O/update_pmc_traffic.json "--only-update --steps 5 (8 updates traced; per-step keys read per UPDATE)"
keep rollout_pmc_fetch_size.csv rollout_pmc_write_size.csv rollout_pmc_mfma_util.csv rollout_pmc_traffic.json update_pmc_fetch_size.csv update_pmc_write_size.csv update_pmc_mfma_util.csv update_pmc_traffic.json
ls -la $O | grep -v "^d"
