"""Kernel timeline of one graph-replayed rollout step from a rocprofv3 `--kernel-trace --output-format csv` trace:
start (us from the step's observation copy), duration, queue, grid, kernel.  python tools/timeline_csv.py <kernel_trace.csv> [step]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_copy_multi" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) * 3 // 4
a, b = idx[k], idx[k + 1]
t0 = int(rows[a]["Start_Timestamp"])
busy = {}
for r in rows[a:b]:
    nm = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    nm = re.sub(r"^void ", "", nm)[:44]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:6.1f}us q{r['Queue_Id']} g{g:>5}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} {nm}")
    busy[r["Queue_Id"]] = busy.get(r["Queue_Id"], 0) + (e - s)
print("step wall us", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3, "per-queue busy us", {q: round(v / 1e3, 1) for q, v in busy.items()})
