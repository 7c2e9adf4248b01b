#!/bin/bash
# Round 5: kernel-trace timeline of one pred-semantics step - launches, durations, gaps between launches per queue
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05n
rm -rf $O && mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace -d $O/trace -- python3 bench.py --steps 20 --warmup 5 --reps 1 --no-cpu-baseline --no-update --no-collect --no-gt-leg > $O/trace.log 2>&1
f=$(find $O/trace -name "*.db" | head -1)
python tools/step_timeline.py $f > $O/step_timeline.txt 2>&1
tail -5 $O/step_timeline.txt
rm -rf $O/trace
