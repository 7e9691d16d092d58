#!/bin/bash
# One GPU call for the reworked training step: the whole GPU suite, tools/bench_train.py in its four modes, and a
# rocprofv3 kernel summary of one eager split-GEMM step.  Output under gpurun_out/<tag>/.
set -u
TAG=${1:-r02_train}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider > "$OUT/gpu_suite.log" 2>&1
echo "suite rc=$?" >> "$OUT/gpu_suite.log"
tail -5 "$OUT/gpu_suite.log"
for MODE in "fp32" "f16x2" "f16x2 --graph" "fp32 --graph"; do
    NAME=$(echo "$MODE" | tr -d ' -')
    timeout 600 python tools/bench_train.py --precision $MODE --steps 5 --warmup 2 > "$OUT/bench_train_$NAME.json" 2> "$OUT/bench_train_$NAME.err"
    echo "bench_train $MODE rc=$?"; cat "$OUT/bench_train_$NAME.json"; tail -3 "$OUT/bench_train_$NAME.err"
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o train --output-format csv -- \
    python "$ROOT/tools/bench_train.py" --precision f16x2 --steps 2 --warmup 1 > "$OUT/bench_train_under_rocprof.json" 2> "$OUT/prof.err"
find "$OUT/prof" -name '*kernel_stats.csv' -exec cp {} "$OUT/train_f16x2_kernel_stats.csv" \;
find "$OUT/prof" -name '*kernel_trace.csv' -delete
head -40 "$OUT/train_f16x2_kernel_stats.csv"
