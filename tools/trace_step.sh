#!/bin/bash
# Per-dispatch kernel durations of one B=64 denoiser forward (tools/pmc_step.py) under rocprofv3 --kernel-trace:
# gpurun_out/<tag>/step_trace.csv (kernel, start, end) + a per-(kernel, order-in-block) summary on stdout.
set -u
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d "$OUT/trace" -o t --output-format csv -- python "$ROOT/tools/pmc_step.py" > "$OUT/trace.log" 2>&1
F=$(find "$OUT/trace" -name '*kernel_trace.csv' | head -1)
python - "$F" "$OUT/step_trace_summary.txt" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ds = [r for r in rows if "ds_" in r["Kernel_Name"]]
# the last forward = the last 19 * 11 + 3 launches before the end; summarise by kernel name and grid size
agg = collections.OrderedDict()
half = ds[len(ds) // 2:]
for r in half:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    key = (name, r.get("Grid_Size", r.get("Grid_Size_X", "?")))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg.setdefault(key, []).append(d)
t0, t1 = int(half[0]["Start_Timestamp"]), int(half[-1]["End_Timestamp"])
out = open(sys.argv[2], "w")
tot = 0.0
for (name, grid), v in agg.items():
    line = "%-70s grid %-9s n=%3d  avg %8.1f us  sum %9.1f us" % (name[:70], grid, len(v), sum(v) / len(v), sum(v))
    tot += sum(v)
    print(line); out.write(line + "\n")
line = "second forward: kernels %.1f us, wall %.1f us (gaps %.1f us)" % (tot, (t1 - t0) / 1e3, (t1 - t0) / 1e3 - tot)
print(line); out.write(line + "\n")
PY
