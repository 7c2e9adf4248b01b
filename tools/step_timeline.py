"""Timeline of ONE graph-replayed step from a rocprofv3 kernel-trace database (rocpd sqlite): per queue, every kernel's start
(us from the step's first kernel), duration and the GAP to the previous kernel's end on the same queue; then, per queue, the
sums - busy time, gap time - and the count of launches.  A step = the kernels between two `k_copy_multi` launches (the
observation copy that opens a step).   python tools/step_timeline.py <results.db> [step index] [--brief]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"^void ", "", name)[:70]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    qcol = [c for c in cols if "queue" in c][0] if any("queue" in c for c in cols) else None
    sel = f"select {namecol}, start, end" + (f", {qcol}" if qcol else ", 0") + " from kernels order by start"
    rows = cur.execute(sel).fetchall()
    idx = [i for i, r in enumerate(rows) if "k_copy_multi" in r[0]]
    args = [a for a in sys.argv[2:] if not a.startswith("--")]
    k = int(args[0]) if args else len(idx) * 3 // 4
    brief = "--brief" in sys.argv
    a, b = idx[k], idx[k + 1]
    t0 = rows[a][1]
    last_end, busy, gaps, count = {}, {}, {}, {}
    for n, s, e, q in rows[a:b]:
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        if not brief:
            print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:6.1f}us gap {gap:5.1f} q{q} {short(n)}")
        busy[q] = busy.get(q, 0) + (e - s) / 1e3
        gaps[q] = gaps.get(q, 0) + max(gap, 0.0)
        count[q] = count.get(q, 0) + 1
        last_end[q] = max(e, last_end.get(q, 0))
    print(f"step wall {(rows[b][1] - t0) / 1e3:.1f} us")
    for q in busy:
        print(f"queue {q}: {count[q]} launches, busy {busy[q]:.1f} us, gaps between launches {gaps[q]:.1f} us (= {gaps[q] / max(count[q] - 1, 1):.2f} per launch)")


if __name__ == "__main__":
    main()
