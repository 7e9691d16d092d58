#!/bin/bash
# The round's profiling recipe in one place (run ON THE GPU BOX, e.g. gpurun -- 'bash tools/profile_round.sh r02').
# Produces under gpurun_out/<tag>/ what profiles/README.md describes; copy the summaries you want judged into profiles/.
#   1. bench.py default line                         -> <tag>_bench_default.json
#   2. the same command under rocprofv3 --stats      -> <tag>_bench_under_rocprof.json + kernel stats csv
#   3. PMC passes over one B=64 denoiser step, one counter set per pass (never combined with trace domains)
#      -> per-pass directories, summarised by tools/pmc_summarize.py
set -u
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp

# PMC passes first: bench.py quotes roofline.traffic from their summary (and only while the kernel sources match it)
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
    NAME=$(echo "$SET" | tr ' ' '_')
    # (the exact form used in round 1: --kernel-trace is allowed next to --pmc, the sys / hip / hsa domains are not)
    timeout 600 rocprofv3 --pmc $SET --kernel-trace -d "$OUT/pmc_step/$NAME" -o p --output-format csv -- \
        python "$ROOT/tools/pmc_step.py" > "$OUT/pmc_${NAME}.log" 2>&1
done
python "$ROOT/tools/pmc_summarize.py" "$OUT/pmc_step" "$OUT/${TAG}_pmc_denoiser_step_b64.csv" \
    "$OUT/${TAG}_pmc_denoiser_step_b64.json" > "$OUT/pmc_summarize.log" 2>&1 || true
cp "$OUT/${TAG}_pmc_denoiser_step_b64.json" "$ROOT/profiles/" 2>/dev/null || true
rm -rf "$OUT/pmc_step"        # the per-pass rocprofv3 directories are hundreds of MB: gpurun merges at most 64 MiB back

timeout 900 python "$ROOT/bench.py" > "$OUT/${TAG}_bench_default.json" 2> "$OUT/bench_default.err"

timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o bench --output-format csv -- \
    python "$ROOT/bench.py" --steps 1 --warmup 1 > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/bench_prof.err"
cp "$OUT/prof/bench_kernel_stats.csv" "$OUT/${TAG}_bench_default_kernel_stats.csv" 2>/dev/null || \
    find "$OUT/prof" -name '*kernel_stats.csv' -exec cp {} "$OUT/${TAG}_bench_default_kernel_stats.csv" \;

rm -rf "$OUT/prof"
ls -la "$OUT"
