"""Per-kernel averages of every counter collected by tools/pmc_sq.sh:  pmc_table.py <dir> [kernel-name substring]"""
import collections
import csv
import glob
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "ds_"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(src + "/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if flt not in k:
        continue
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("    %-28s %14.4g   (%d dispatches)" % (c, sum(v) / len(v), len(v)))
