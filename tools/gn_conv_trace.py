"""Per-launch durations of the k_gn_conv chain from a rocprofv3 --kernel-trace CSV (one step's worth, averaged over
the last steps).  usage: python tools/gn_conv_trace.py <kernel_trace.csv> [launches per step = 53]"""
import csv
import sys

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = int(sys.argv[2]) if len(sys.argv) > 2 else 53
gn = [r for r in rows if "k_gn_conv" in r["Kernel_Name"]]
steps = min(20, len(gn) // per - 1)
seq = gn[-per * steps:]
d = np.array([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seq]).reshape(steps, per) / 1e3
st = np.array([int(r["Start_Timestamp"]) for r in seq]).reshape(steps, per) / 1e3
en = np.array([int(r["End_Timestamp"]) for r in seq]).reshape(steps, per) / 1e3
for i in range(per):
    r = seq[i]
    gap = np.median(st[:, i] - en[:, i - 1]) if i else 0.0
    print(f"{i:3d} {r['Kernel_Name'].split('k_gn_conv')[1][:9]} S={r['Grid_Size_Y']} lds={r['LDS_Block_Size']:>6} vgpr={r['VGPR_Count']:>3} "
          f"dur {np.median(d[:, i]):6.1f}  gap-before {gap:6.1f}")
print("sum of durations %.1f us, chain span %.1f us" % (np.median(d, 0).sum(), np.median(en[:, -1] - st[:, 0])))
