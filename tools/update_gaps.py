"""Gaps between consecutive kernels of one DAgger update from a rocprofv3 `--kernel-trace --output-format csv` trace of
`bench.py --only-update`: python tools/update_gaps.py <kernel_trace.csv>"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# one update = from a k_map_features* launch to the next
idx = [i for i, r in enumerate(rows) if "k_map_features" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
end = t0
busy = 0
gaps = []
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > end:
        gaps.append(((s - end) / 1e3, (s - t0) / 1e3, re.sub(r"\(.*", "", r["Kernel_Name"])[:40]))
    busy += max(0, e - max(s, end))
    end = max(end, e)
print(f"update wall {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, GPU busy (union) {busy / 1e3:.1f} us, "
      f"{len(gaps)} gaps totalling {sum(g[0] for g in gaps):.1f} us")
for g in sorted(gaps, reverse=True)[:25]:
    print(f"  gap {g[0]:7.1f} us before {g[2]:40s} at {g[1]:9.1f} us")
