#!/bin/bash
# One GPU call for the training step: its GPU tests (or the whole suite with FULL=1), tools/bench_train.py in several
# modes, and a rocprofv3 kernel summary of one eager split-GEMM step.  Output under gpurun_out/<tag>/.
set -u
TAG=${1:-r02_train}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
if [ "${FULL:-0}" = "1" ]; then SEL="tests"; else SEL="tests/test_hip_train_kernels.py"; fi
timeout 900 python -m pytest $SEL -m gpu -q -s -p no:cacheprovider > "$OUT/gpu_tests.log" 2>&1
echo "tests rc=$?" >> "$OUT/gpu_tests.log"
grep -n "rel err\|worst\|iter \|FAILED\|passed\|failed\|rc=" "$OUT/gpu_tests.log" | tail -40
while IFS= read -r MODE; do
    [ -z "$MODE" ] && continue
    NAME=$(echo "$MODE" | tr -d ' -')
    timeout 600 python tools/bench_train.py --precision $MODE --steps 5 --warmup 2 > "$OUT/bench_train_$NAME.json" 2> "$OUT/bench_train_$NAME.err"
    echo "bench_train $MODE rc=$?"; cut -c1-400 "$OUT/bench_train_$NAME.json"; grep -v amdgpu.ids "$OUT/bench_train_$NAME.err" | tail -3
done <<LIST
fp32
f16x2
f16x2 --graph
f16x2 --graph --attention composed
LIST
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o train --output-format csv -- \
    python "$ROOT/tools/bench_train.py" --precision f16x2 --steps 2 --warmup 1 > "$OUT/bench_train_under_rocprof.json" 2> "$OUT/prof.err"
find "$OUT/prof" -name '*kernel_stats.csv' -exec cp {} "$OUT/train_f16x2_kernel_stats.csv" \;
find "$OUT/prof" -name '*kernel_trace.csv' -delete
cut -c1-150 "$OUT/train_f16x2_kernel_stats.csv" | head -24
