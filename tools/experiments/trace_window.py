#!/usr/bin/env python3
"""Print the kernels of a time window of a rocprofv3 kernel-trace CSV: start/end (us from window start), queue, name."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0][:28]))
rows.sort()
frac = float(sys.argv[2]); width = float(sys.argv[3]) * 1e6
t0 = rows[0][0] + (rows[-1][0] - rows[0][0]) * frac
qs = {}
for s, e, q, n in rows:
    if s < t0 or s > t0 + width: continue
    col = qs.setdefault(q, len(qs))
    print(f"{(s-t0)/1e3:9.1f} {(e-t0)/1e3:9.1f} {(e-s)/1e3:8.1f}  " + "                                  " * col + n)
