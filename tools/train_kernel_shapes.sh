#!/bin/bash
# Per (kernel, grid size) calls and average duration of the captured training iteration (run ON THE GPU BOX): the per-kernel
# statistics of tools/train_profile.sh cannot tell the launches of one kernel apart (which GEMM shape costs what in situ).
# usage: tools/train_kernel_shapes.sh <out dir> [bench_train args]
set -u
OUT=$(mkdir -p "$1" && cd "$1" && pwd)
shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d "$OUT/prof" -o tr --output-format csv -- python "$ROOT/tools/bench_train.py" --precision f16x2 --graph --steps 20 --warmup 2 "$@" > "$OUT/bench_train_trace.log" 2>&1
python - "$OUT" <<'PY' > "$OUT/train_kernel_shapes.txt"
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/prof/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    k = (r["Kernel_Name"][:70], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""))
    a = agg[k]
    a[0] += 1
    a[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
tot = sum(a[1] for a in agg.values())
print("total %.1f ms in %d launches" % (tot / 1e6, sum(a[0] for a in agg.values())))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:90]:
    print("%8.2f ms %6d calls %8.1f us  grid %8s x %-4s %s" % (a[1] / 1e6, a[0], a[1] / a[0] / 1e3, k[1], k[2], k[0]))
PY
rm -rf "$OUT/prof"
head -70 "$OUT/train_kernel_shapes.txt" | cut -c1-150
