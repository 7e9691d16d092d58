#!/bin/bash
# Per-kernel time of the 64-clip SpecVQGAN decode + MelGAN vocode (run ON THE GPU BOX): rocprofv3 --kernel-trace --stats over
# tools/decode_time.py; prints the top kernels.  usage: tools/decode_profile.sh <out dir>
set -u
OUT=$1
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o dec --output-format csv -- python "$ROOT/tools/decode_time.py" > "$OUT/decode_time.log" 2>&1
find "$OUT/prof" -name '*kernel_stats.csv' -exec cp {} "$OUT/decode_kernel_stats.csv" \;
rm -rf "$OUT/prof"
python - "$OUT/decode_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print("%-100s calls %5s  total %8.2f ms  avg %8.1f us  %5.1f%%" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
          float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
