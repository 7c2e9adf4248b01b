"""Summarise the memory copies of a rocprofv3 rocpd database (`--memory-copy-trace`): count / bytes by direction and
size.  python tools/memcopy_stats.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
cands = [t for t in tables if "memory_cop" in t.lower()]
print("tables:", cands)
for t in cands:
    cols = [r[1] for r in cur.execute(f"pragma table_info({t})")]
    print(t, cols)
    try:
        rows = cur.execute(f"select * from {t} limit 3").fetchall()
        for r in rows:
            print("   ", r)
        namecol = next((c for c in cols if c in ("name", "kind", "direction")), None)
        sizecol = next((c for c in cols if "size" in c or "bytes" in c), None)
        if namecol and sizecol:
            for r in cur.execute(f"select {namecol}, {sizecol}, count(*) from {t} group by {namecol}, {sizecol} order by count(*) desc limit 25"):
                print("   ", r)
    except Exception as e:  # noqa: BLE001
        print("   ", e)
