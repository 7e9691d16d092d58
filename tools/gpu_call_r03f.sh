#!/bin/bash
# Round-3 GPU call F: half-tile program -- GPU suite, then BASELINE configs[1] (batch 32, transformer only) and the default line.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-r03f}
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests -m gpu -q -x > "$OUT/gpu_suite.log" 2>&1
tail -12 "$OUT/gpu_suite.log"
timeout 900 python bench.py --transformer-only --batch 32 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_cfg1_b32_transformer_only.json" 2> "$OUT/bench_cfg1.err"
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python - <<PY
import json
for f in ("bench_cfg1_b32_transformer_only", "bench_default"):
    try:
        d = json.loads(open("$OUT/%s.json" % f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], r["frac"], r["avg_launch_us"], {k: (v["launches"], v["avg_launch_us"], v["tflops"]) for k, v in r["all_gemm_tiles"].items()}, r["all_gemm_tflops"])
    except Exception as e:
        print(f, "failed", e)
PY
tail -3 "$OUT/bench_cfg1.err"
