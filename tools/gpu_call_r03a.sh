#!/bin/bash
# Round-3 GPU call A: sustained ablation ladder of the dominant GEMM with power / clock telemetry, the same at 32 tiles
# (does the epilogue get faster when only 32 CUs run it?), operand-data dependence, and board telemetry under bench.py.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r03a
mkdir -p "$OUT"
cd "$ROOT"
rocm-smi --showpower --showclocks --showmaxpower --showtemp > "$OUT/smi_idle.txt" 2>&1
timeout 300 ./tools/probe/probe_ceiling 0.5 64 > "$OUT/ceiling_b64.txt" 2>&1
timeout 200 ./tools/probe/probe_ceiling 0.3 8 > "$OUT/ceiling_b8.txt" 2>&1
PROBE_ZERO_LO=1 timeout 200 ./tools/probe/probe_ceiling 0.3 64 > "$OUT/ceiling_b64_zero_lo.txt" 2>&1
PROBE_ZERO_LO=2 timeout 200 ./tools/probe/probe_ceiling 0.3 64 > "$OUT/ceiling_b64_zero_all.txt" 2>&1
( while true; do rocm-smi --showpower --showclocks --csv 2>/dev/null | tr '\n' ' '; echo; sleep 0.25; done ) > "$OUT/smi_during_bench.txt" &
SMI=$!
timeout 900 python bench.py --steps 4 --warmup 1 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
kill $SMI
tail -c 1500 "$OUT/ceiling_b64.txt"
