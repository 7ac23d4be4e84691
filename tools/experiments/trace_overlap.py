#!/usr/bin/env python3
"""Overlap report from a rocprofv3 --kernel-trace CSV: per kernel name the summed duration, the wall span of the
trace, the busy time (union of intervals) and how much of each kernel's time ran alongside another kernel."""
import csv, sys, collections

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5        # ignore the first part (warm-up, setup)
t0, t1 = rows[0][0], max(r[1] for r in rows)
cut = t0 + (t1 - t0) * skip
rows = [r for r in rows if r[0] >= cut]
t0, t1 = rows[0][0], max(r[1] for r in rows)
ev = []
for s, e, n in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = 0; depth = 0; last = t0; multi = 0
for t, d in ev:
    if depth > 0: busy += t - last
    if depth > 1: multi += t - last
    depth += d; last = t
tot = collections.Counter(); cnt = collections.Counter(); ov = collections.Counter()
ends = sorted(rows, key=lambda r: r[0])
for i, (s, e, n) in enumerate(ends):
    tot[n] += e - s; cnt[n] += 1
    o = 0
    for j in range(max(0, i - 40), min(len(ends), i + 40)):
        if j == i: continue
        s2, e2, _ = ends[j]
        o = max(o, 0) + max(0, min(e, e2) - max(s, s2))
    ov[n] += min(o, e - s)
print(f"span {1e-6*(t1-t0):.2f} ms  busy {1e-6*busy:.2f} ms  >=2 kernels {1e-6*multi:.2f} ms  sum of durations {1e-6*sum(tot.values()):.2f} ms")
for n, v in tot.most_common(24):
    print(f"{n[:40]:40s} n={cnt[n]:5d} total {1e-6*v:8.2f} ms  avg {1e-3*v/cnt[n]:8.1f} us  overlapped {100.0*ov[n]/v:5.1f}%")
