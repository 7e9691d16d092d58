#!/bin/bash
# Round-3 call L: GPU suite on the residual-ahead ROW epilogue, the trained-like numbers with their prints, and the same-run
# A/B of the per-sample GEMM against the previous revision (tools/probe/prev) on the four shapes of a block.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r03l
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/gpu_suite.log" 2>&1
tail -3 "$OUT/gpu_suite.log"
timeout 300 python -m pytest tests/test_hip_trained_like.py -m gpu -q -s 2>&1 | grep -E "trained-like|K=512|passed|failed" > "$OUT/trained_like_measured.txt"
cat "$OUT/trained_like_measured.txt"
PROBE_VARIANTS=0,13,12,3 timeout 300 ./tools/probe/probe_ceiling 0.4 64 > "$OUT/ps_kernel_row_epilogue_ab.txt" 2>&1
PROBE_VARIANTS=13,0,12,3 timeout 300 ./tools/probe/probe_ceiling 0.4 64 >> "$OUT/ps_kernel_row_epilogue_ab.txt" 2>&1
cat "$OUT/ps_kernel_row_epilogue_ab.txt"
