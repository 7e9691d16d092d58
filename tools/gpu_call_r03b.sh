#!/bin/bash
# Round-3 GPU call B: what the epilogue's time is made of (LDS staging vs global traffic), SQ counters of the main-loop
# variants, and the GPU test suite with the new in-kernel-noise entries.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r03b
mkdir -p "$OUT"
cd "$ROOT"
PROBE_VARIANTS=0,1,2,3 timeout 300 ./tools/probe/probe_ceiling 0.4 64 > "$OUT/epilogue_split_b64.txt" 2>&1
PROBE_VARIANTS=0,1,2,3 timeout 300 ./tools/probe/probe_ceiling 0.3 8 > "$OUT/epilogue_split_b8.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/rocprof_counters.txt" 2>&1
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"; do
    NAME=$(echo "$SET" | tr ' ' '_' | cut -c1-40)
    PROBE_VARIANTS=0,3,4,5 timeout 300 rocprofv3 --pmc $SET --kernel-trace -d "$OUT/pmc/$NAME" -o p --output-format csv -- \
        "$ROOT/tools/probe/probe_ceiling" 0.03 64 > "$OUT/pmc_$NAME.log" 2>&1
done
cd "$ROOT"
timeout 1200 python -m pytest tests -m gpu -x -q > "$OUT/gpu_suite.log" 2>&1
tail -5 "$OUT/gpu_suite.log"
cat "$OUT/epilogue_split_b64.txt" | tail -30
