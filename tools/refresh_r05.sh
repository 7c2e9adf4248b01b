#!/bin/bash
# Round-5 profile set (run on the GPU box through gpurun; outputs under gpurun_out/r05/, copied to profiles/r05_* by hand).
# Kernel-trace passes and PMC passes are separate runs; every run is bounded by a timeout.  bench.py's headline is
# configs[2] (pred semantics, 8 envs) since round 4; --gt-semantics makes configs[1] (4 envs) the headline of a run.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05
rm -rf $O && mkdir -p $O
ONLY="--no-cpu-baseline --no-update --no-collect"
PRED="$ONLY --no-gt-leg"
GT="$ONLY --gt-semantics --no-pred-leg"
run() {  # name, rocprof args..., -- program args
  name=$1; shift
  timeout -k 10 ${T:-170} rocprofv3 "$@" > $O/$name.log 2>&1
  echo "$name rc=$?"
}
stats() { f=$(find $O/$1 -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_stats.py $f $O/$2 > /dev/null; }
pmc() { f=$(find $O/$1 -name "*.db" | head -1); [ -n "$f" ] && python tools/pmc_stats.py $f $O/$2 > /dev/null; }
# --- un-profiled micro-benchmarks and the driver's line first (a clean GPU) ---
timeout 200 python tools/depth_net_phases.py 4 > $O/depth_net_phases_N4.txt 2>&1
timeout 100 python tools/depth_net_phases.py 8 2>&1 | grep "per launch\|sum of" > $O/depth_net_phases_N8.txt
timeout 600 python bench.py > $O/bench_full.json 2> $O/bench_full.err
timeout 300 python bench.py --gt-semantics --envs 8 --no-update --no-collect --no-pred-leg --no-cpu-baseline > $O/bench_gt_B8.json 2> $O/bench_gt_B8.err
IVLN_REDNET_PLAN=0 timeout 200 python tools/gemm_shapes.py rednet > $O/rednet_B8_gemm_shapes.txt 2>&1
# the split-bf16 conv against the fp32 MFMA kernels: time, error against float64, and where its workgroups spend their time
timeout 300 python tools/conv_bf3_probe.py all > $O/conv_bf3_probe.txt 2>&1
timeout 300 python tools/conv_bf3_phases.py > $O/conv_bf3_phases.txt 2>&1
IVLN_SPLIT_BF16=0 timeout 300 python bench.py --no-cpu-baseline --no-collect --reps 3 > $O/bench_fp32_only.json 2> $O/bench_fp32_only.err
# A/B of this round's kernels inside ONE call (boxes differ by ~6 %): the pred-semantics step without each of them
P5="--no-update --no-collect --no-gt-leg --no-cpu-baseline --reps 3"
for v in "IVLN_X=1" "IVLN_BF3_KS=0" "IVLN_BF3_1X1_KS=0" "IVLN_BF3_FUSE=0" "IVLN_BF3_NOSPLIT4=0" "IVLN_BF3_KS_TN=2" "IVLN_CACHE_INSTRUCTION=0" "IVLN_X=1"; do
  env $v timeout 300 python bench.py $P5 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$v /"
done > $O/predsem_ab.txt 2>&1
timeout 300 python tools/update_torch_ops.py > $O/update_torch_ops.txt 2>&1
# the wave-split kernels' per-wave phases, the pipe's issue rate, and every candidate kernel / tile per RedNet shape as replayed graphs
timeout 600 python tools/conv_bf3_ks_phases.py 2>&1 | grep -v amdgpu.ids > $O/conv_bf3_ks_phases.txt
timeout 60 tools/mfma_rate > $O/mfma_rate.txt 2>&1
timeout 900 python tools/conv_cfg_sweep.py 2>&1 | grep -v amdgpu.ids > $O/conv_cfg_sweep.txt
# where the split replay's time goes (end of each graph, per step), with and without the per-episode instruction cache
for B in 4 8; do
  for m in 1 0; do echo "== gt envs $B IVLN_CACHE_INSTRUCTION=$m"; IVLN_CACHE_INSTRUCTION=$m timeout 200 python tools/split_probe.py gt $B 2>&1 | tail -2; done
done > $O/split_probe.txt 2>&1
echo "== pred envs 8" >> $O/split_probe.txt; timeout 200 python tools/split_probe.py pred 8 2>&1 | tail -2 >> $O/split_probe.txt
# --- kernel traces ---
run predsem --kernel-trace -d $O/predsem -- python3 bench.py --steps 50 --warmup 5 --reps 1 $PRED;               stats predsem predsem_B8_graph_kernel_stats.csv
f=$(find $O/predsem -name "*.db" | head -1); [ -n "$f" ] && python tools/step_timeline.py $f > $O/predsem_B8_step_timeline.txt 2>&1
run graph   --kernel-trace -d $O/graph   -- python3 bench.py --steps 200 --warmup 20 --reps 1 $GT;               stats graph rollout_graph_kernel_stats.csv
run eager   --kernel-trace -d $O/eager   -- python3 bench.py --steps 200 --warmup 20 --reps 1 $GT --no-graph;    stats eager rollout_eager_kernel_stats.csv
run update  --kernel-trace -d $O/update  -- python3 bench.py --only-update --steps 5;                            stats update update_T64N8_kernel_stats.csv
run rednet  --kernel-trace -d $O/rednet  -- python3 tools/bench_components.py rednet;                           stats rednet rednet_B8_kernel_stats.csv
grep -h '"metric"' $O/predsem.log | tail -1 > $O/bench_predsem_B8.json
grep -h '"metric"' $O/graph.log | tail -1 > $O/bench_gt_graph.json
# --- PMC passes: one counter per run, kernel trace only ---
for c in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | tr A-Z a-z)
  run p_$n --kernel-trace --pmc $c -d $O/p_$n -- python3 bench.py --steps 10 --warmup 2 --reps 1 $PRED --no-graph;  pmc p_$n predsem_B8_pmc_$n.csv
  run r_$n --kernel-trace --pmc $c -d $O/r_$n -- python3 bench.py --steps 20 --warmup 5 --reps 1 $GT --no-graph;    pmc r_$n rollout_pmc_$n.csv
  run u_$n --kernel-trace --pmc $c -d $O/u_$n -- python3 bench.py --only-update --steps 5;                          pmc u_$n update_pmc_$n.csv
done
run pmfma --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $O/p_mfma -- python3 bench.py --steps 10 --warmup 2 --reps 1 $PRED --no-graph; pmc p_mfma predsem_B8_pmc_mfma_util.csv
run rmfma --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $O/r_mfma -- python3 bench.py --steps 20 --warmup 5 --reps 1 $GT --no-graph;  pmc r_mfma rollout_pmc_mfma_util.csv
run umfma --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $O/u_mfma -- python3 bench.py --only-update --steps 5;                         pmc u_mfma update_pmc_mfma_util.csv
find $O -name "*.db" -delete
# steps traced per pass: gt = warm-up 5 + timed 20 + instrumented roofline passes 20 + 20 (mapper) = 65; pred = 2 + 10 + 6 = 18; update = 2 + 5 + 1 = 8
python tools/pmc_traffic.py $O/rollout_pmc_fetch_size.csv $O/rollout_pmc_write_size.csv 65 $O/rollout_pmc_traffic.json "--gt-semantics --envs 4 --steps 20 --warmup 5 --reps 1 (depth encoder = the persistent launch, eager)" 45
python tools/pmc_traffic.py $O/predsem_B8_pmc_fetch_size.csv $O/predsem_B8_pmc_write_size.csv 18 $O/predsem_B8_pmc_traffic.json "--pred-envs 8 --steps 10 --warmup 2 --reps 1"
python tools/pmc_traffic.py $O/update_pmc_fetch_size.csv $O/update_pmc_write_size.csv 8 $O/update_pmc_traffic.json "--only-update --steps 5 (8 updates traced; per-step keys read per UPDATE)"
ls -la $O | grep -v "^d"
