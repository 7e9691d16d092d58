#!/bin/bash
# Per-kernel time of the training iteration (BASELINE configs[4] on one GPU, run ON THE GPU BOX): rocprofv3 --kernel-trace --stats over
# tools/bench_train.py (eager f16x2 iterations; extra arguments go to bench_train.py, e.g. --graph = the captured iteration, whose
# kernels rocprofv3 reports per replay); prints the top kernels.  usage: tools/train_profile.sh <out dir> [bench_train args]
set -u
OUT=$(mkdir -p "$1" && cd "$1" && pwd)       # absolute: rocprofv3 runs from /tmp
shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o tr --output-format csv -- python "$ROOT/tools/bench_train.py" --precision f16x2 --steps 4 --warmup 2 "$@" > "$OUT/bench_train.log" 2>&1
find "$OUT/prof" -name '*kernel_stats.csv' -exec cp {} "$OUT/train_kernel_stats.csv" \;
rm -rf "$OUT/prof"
tail -1 "$OUT/bench_train.log" | cut -c1-400
python - "$OUT/train_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms" % (tot / 1e6))
for r in rows[:30]:
    print("%-110s calls %5s  total %8.2f ms  avg %8.1f us  %5.1f%%" % (r["Name"][:110], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
          float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
