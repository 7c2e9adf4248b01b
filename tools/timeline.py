"""Print the kernel timeline of one rollout step from a rocprofv3 rocpd database:
python tools/timeline.py <results.db> [step_index]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end,queue_id,grid_x,grid_y,grid_z from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "k_local_unproject" in r[0]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
a, b = idx[k], idx[k + 1]
t0 = rows[a][1]
busy = {}
for r in rows[a:b]:
    nm = re.sub(r"\(anonymous namespace\)::", "", r[0])
    nm = re.sub(r"^void ", "", nm)[:40]
    print(f"{(r[1]-t0)/1e3:8.1f} {(r[2]-r[1])/1e3:6.1f}us q{r[3]} g{(r[4] or 0)//256:>5}x{r[5]}x{r[6]} {nm}")
    busy[r[3]] = busy.get(r[3], 0) + (r[2] - r[1])
print("step wall us", (rows[b][1] - t0) / 1e3, "per-queue busy us", {q: v / 1e3 for q, v in busy.items()})
