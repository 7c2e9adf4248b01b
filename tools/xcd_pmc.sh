#!/bin/bash
# HBM fetch bytes of the update step's kernels with and without the XCD-aware workgroup remap
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for v in on off; do
  O=gpurun_out/xcd_$v; rm -rf $O
  if [ $v = off ]; then export IVLN_NO_XCD_REMAP=1; else unset IVLN_NO_XCD_REMAP; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -- python3 tools/bench_components.py update > $O.log 2>&1
  echo "$v rc=$?"
  f=$(find $O -name "*.db" | head -1)
  python tools/pmc_stats.py $f gpurun_out/xcd_${v}_fetch.csv > /dev/null
  find $O -name "*.db" -delete
done
python - <<'PY'
import csv
a={r['Name']:r for r in csv.DictReader(open('gpurun_out/xcd_on_fetch.csv'))}
b={r['Name']:r for r in csv.DictReader(open('gpurun_out/xcd_off_fetch.csv'))}
ta=sum(float(r['Total']) for r in a.values()); tb=sum(float(r['Total']) for r in b.values())
print(f"total FETCH_SIZE KiB: remap {ta:.0f}  identity {tb:.0f}  ratio {ta/tb:.3f}")
for n,r in sorted(b.items(), key=lambda kv:-float(kv[1]['Total']))[:14]:
    if n in a: print(f"{n[:70]:70s} identity {float(r['PerLaunch']):10.0f}  remap {float(a[n]['PerLaunch']):10.0f}  x{float(a[n]['PerLaunch'])/max(1,float(r['PerLaunch'])):.2f}")
PY
