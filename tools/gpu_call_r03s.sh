#!/bin/bash
# Round-3 call S: GPU suite on the pack / unzip micro-changes of the per-sample epilogue (v_cvt_pk_f16_f32, v_perm_b32) and
# the same-run A/B against the previous revision (tools/probe/prev).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r03s
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/gpu_suite.log" 2>&1
tail -3 "$OUT/gpu_suite.log"
PROBE_VARIANTS=0,13 timeout 300 ./tools/probe/probe_ceiling 0.4 64 > "$OUT/ps_kernel_pack_unzip_ab.txt" 2>&1
PROBE_VARIANTS=13,0 timeout 300 ./tools/probe/probe_ceiling 0.4 64 >> "$OUT/ps_kernel_pack_unzip_ab.txt" 2>&1
grep -v rocm_smi "$OUT/ps_kernel_pack_unzip_ab.txt"
