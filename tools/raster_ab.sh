#!/bin/bash
# Raster A/B of the per-sample GEMM program (run ON THE GPU BOX): PS_GM = samples per raster group of gemm_f16x2_ps.hip (product: 4).
# Variant libraries are built in the build container from a sed-patched COPY of the source (the product file is not touched):
#   for gm in 8 2; do sed "s/#define PS_GM 4 /#define PS_GM $gm /" csrc/gemm_f16x2_ps.hip > /tmp/ps_gm$gm.hip; hipcc ... -> gpurun_ab_gm$gm.so
# For each library: bench.py (clips/s, GEMM us per launch) and the FETCH_SIZE / WRITE_SIZE passes over tools/pmc_step.py.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-r05r}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for V in product gm8 gm2; do
    if [ $V = product ]; then unset DIFFSOUND_LIB; else export DIFFSOUND_LIB=$ROOT/gpurun_ab_$V.so; fi
    python "$ROOT/bench.py" --steps 2 --warmup 1 --no-train-leg --no-cpu-baseline > "$OUT/bench_$V.json" 2> "$OUT/bench_$V.err"
    for SET in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
        NAME=$(echo "$SET" | tr ' ' '_')
        timeout 300 rocprofv3 --pmc $SET --kernel-trace -d "$OUT/pmc_$V/$NAME" -o p --output-format csv -- python "$ROOT/tools/pmc_step.py" > "$OUT/pmc_${V}_$NAME.log" 2>&1
    done
    python "$ROOT/tools/pmc_summarize.py" "$OUT/pmc_$V" "$OUT/pmc_$V.csv" "$OUT/pmc_$V.json" > "$OUT/pmc_summarize_$V.log" 2>&1
    rm -rf "$OUT/pmc_$V"
done
python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
for v in ("product", "gm8", "gm2"):
    b = json.load(open("%s/bench_%s.json" % (out, v)))
    p = json.load(open("%s/pmc_%s.json" % (out, v)))
    k = {n: r for n, r in p.items() if n.startswith("ds_gemm_f16x2_ps_kernel")}
    nd = sum(r["dispatches"] for r in k.values())
    rd = sum(r["hbm_read_MB_per_launch"] * r["dispatches"] for r in k.values()) / nd
    wr = sum(r["hbm_write_MB_per_launch"] * r["dispatches"] for r in k.values()) / nd
    l2 = sum(r["l2_hit_rate"] * r["dispatches"] for r in k.values()) / nd
    print("%-8s %.2f clips/s  GEMM %.1f us per launch (frac %.4f)   fabric read %.0f MB + write %.0f MB per launch, L2 hit %.3f"
          % (v, b["value"], b["roofline"]["avg_launch_us"], b["roofline"]["frac"], rd, wr, l2))
PY
