#!/bin/bash
# Round-3 closing measurements: tools/profile_round.sh (PMC passes over the sampling step, the default bench line, the
# same command under rocprofv3 --stats), the side lines (configs[1] at batch 32, the K = 512 leg), smoke().
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r03k}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
bash tools/profile_round.sh "$TAG" > "$OUT/profile_round.log" 2>&1
timeout 900 python bench.py --transformer-only --batch 32 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/${TAG}_bench_cfg1_b32_transformer_only.json" 2> "$OUT/cfg1.err"
timeout 900 python bench.py --codes 512 --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/${TAG}_bench_cfg3_k512_b64_1gpu.json" 2> "$OUT/cfg3.err"
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/${TAG}_smoke.log" 2>&1
python - <<PY
import json
for f in ("${TAG}_bench_default", "${TAG}_bench_under_rocprof", "${TAG}_bench_cfg1_b32_transformer_only", "${TAG}_bench_cfg3_k512_b64_1gpu"):
    try:
        d = json.loads(open("$OUT/%s.json" % f).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        print(f, d["value"], d["ms_per_step"], r.get("frac"), r.get("avg_launch_us"), r.get("traffic"), (r.get("traffic_note") or "")[:160], d.get("cpu_baseline"))
    except Exception as e:
        print(f, "failed", e)
PY
tail -2 "$OUT/${TAG}_smoke.log"
head -5 "$OUT/${TAG}_bench_default_kernel_stats.csv" 2>/dev/null | cut -c1-200
