#!/bin/bash
# Round 5: what bounds the 1x1 wave-split kernels' K loop - per-section shader cycles
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05j
mkdir -p $O
timeout 600 python tools/conv_bf3_ks_phases.py 2>&1 | grep -v amdgpu.ids | grep -E "wt|ks|shape" | head -14 | cut -c1-300 > $O/ks_sections.txt
cat $O/ks_sections.txt
