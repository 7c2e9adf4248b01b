"""Per-kernel sums of one PMC counter from a rocprofv3 rocpd database (`--pmc X --kernel-trace`):
python tools/pmc_stats.py <results.db> [out.csv]
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB-like units of the counter's expression; the raw
value is printed per launch together with the kernel's launch count so bytes/launch can be derived."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"^void ", "", name)[:90]


db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
agg = {}
for k, c, v in rows:
    a = agg.setdefault((short(k), c), [0, 0.0])
    a[0] += 1
    a[1] += v
lines = ["Name,Counter,Launches,Total,PerLaunch"]
for (k, c), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append(f"\"{k}\",{c},{a[0]},{a[1]:.1f},{a[1] / a[0]:.2f}")
out = "\n".join(lines)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
print(out)
