"""Two-queue timeline of one rollout step from a rocprofv3 rocpd database: python tools/timeline2.py <db> [step]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end,queue_id,grid_x,grid_y,grid_z from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "k_copy_multi" in r[0] or "k_local_unproject" in r[0] and False]
if not idx:
    idx = [i for i, r in enumerate(rows) if "k_local_unproject" in r[0]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
a, b = idx[k], idx[k + 1]
t0 = rows[a][1]
busy = {}
last_end = {}
for r in rows[a:b]:
    nm = re.sub(r"\(anonymous namespace\)::", "", r[0])
    nm = re.sub(r"^void ", "", nm)[:34]
    q = r[3]
    gap = (r[1] - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = r[2]
    print(f"{(r[1]-t0)/1e3:8.1f} {(r[2]-r[1])/1e3:6.1f}us gap{gap:6.1f} q{q} {nm}")
    busy[q] = busy.get(q, 0) + (r[2] - r[1])
print("step wall us", (rows[b][1] - t0) / 1e3, "per-queue busy us", {q: round(v / 1e3, 1) for q, v in busy.items()})
