#!/bin/bash
# The three SQ passes of tools/pmc_sq.sh only (no TCC / FETCH passes):  tools/pmc_sq_short.sh <out dir> <command...>
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
    NAME=$(echo "$SET" | tr ' ' '_' | cut -c1-40)
    timeout 300 rocprofv3 --pmc $SET --kernel-trace -d "$OUT/$NAME" -o p --output-format csv -- "$@" > "$OUT/$NAME.log" 2>&1
done
