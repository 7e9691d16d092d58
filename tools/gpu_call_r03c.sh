#!/bin/bash
# Round-3 GPU call C: the pipelined epilogue -- GPU test suite first (bit-identity of the per-sample program against the
# 4-wave programs for every store family), then the sustained probe and the bench line.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-r03c}
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/gpu_suite.log" 2>&1
tail -15 "$OUT/gpu_suite.log"
PROBE_VARIANTS=0,3 timeout 300 ./tools/probe/probe_ceiling 0.4 64 > "$OUT/ceiling_b64.txt" 2>&1
PROBE_VARIANTS=0,3 timeout 300 ./tools/probe/probe_ceiling 0.3 8 > "$OUT/ceiling_b8.txt" 2>&1
grep -v rocm_smi "$OUT/ceiling_b64.txt"
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python -c "
import json,sys
d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
