"""Summarise rocprofv3 --pmc CSV output per kernel name: mean over the launches and the value of the largest launch
(the launches of one kernel differ by orders of magnitude where a stage runs once per pyramid octave).
usage: pmc_summary.py <dir> [kernel name filter]"""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} mean {sum(v)/len(v):16.1f}  max {max(v):16.1f}  n={len(v)}")
