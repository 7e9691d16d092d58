#!/bin/bash
# Round-3 call O: SQ counters of the final per-sample GEMM and its ablations (product / no epilogue / no epilogue + no DMA /
# MFMA + barriers only), for the "what is left" accounting of DESIGN.md section 3.  Counters only (no trace domains besides
# --kernel-trace), one pass per counter set.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r03o
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"; do
    NAME=$(echo "$SET" | tr ' ' '_' | cut -c1-40)
    PROBE_VARIANTS=0,3,4,5 timeout 300 rocprofv3 --pmc $SET --kernel-trace -d "$OUT/pmc/$NAME" -o p --output-format csv -- \
        "$ROOT/tools/probe/probe_ceiling" 0.03 64 > "$OUT/pmc_$NAME.log" 2>&1
done
ls -R "$OUT/pmc" | head -20
