"""Aggregate the kernel launches between two marker kernels of a rocprofv3 kernel trace:
    python tools/trace_region.py <kernel_trace.csv> <start marker> <end marker>
Region = after the LAST launch matching <start marker> that still has a launch matching <end marker> behind it, up to (and
including) the first such launch.  Prints wall time, the sum of kernel durations, idle time, and the launches by kernel name.
(Used for: what runs between the end of one batch's vocode and the first sampling step of the next.)"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, n in enumerate(names) if sys.argv[2] in n]
ends = [i for i, n in enumerate(names) if sys.argv[3] in n]
region = None
for s in reversed(starts):
    after = [e for e in ends if e > s]
    if after:
        region = (s + 1, after[0])
        break
if region is None:
    sys.exit("markers not found")
seg = rows[region[0]:region[1] + 1]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
agg = collections.OrderedDict()
busy, last_end = 0, t0
for r in seg:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:100]
    c = agg.setdefault(k, [0, 0.0])
    c[0] += 1
    c[1] += (b - a) / 1e3
    if b > last_end:
        busy += b - max(a, last_end)
        last_end = b
print("region: %d launches, wall %.1f us, kernel sum %.1f us, busy %.1f us, idle %.1f us"
      % (len(seg), (t1 - t0) / 1e3, sum(v[1] for v in agg.values()), busy / 1e3, (t1 - t0 - busy) / 1e3))
for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-100s n=%4d  sum %9.1f us  avg %8.1f us" % (k, n, us, us / n))
