#!/bin/bash
# tools/probe/probe_ceiling.hip includes the per-sample GEMM once per PS_ABLATE value: build it against a scratch copy of csrc/
# with tools/probe/ps_probe.patch applied (the product source carries no ablation switches).
set -eu
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SCRATCH=$(mktemp -d /tmp/ds_probe_src.XXXXXX)
cp "$ROOT"/text-to-sound-synthesis_amd/csrc/*.hip "$ROOT"/text-to-sound-synthesis_amd/csrc/*.inc "$ROOT"/text-to-sound-synthesis_amd/csrc/*.h "$SCRATCH"/
patch -s -d "$SCRATCH" -p1 < "$ROOT/tools/probe/ps_probe.patch"
sed "s#../../text-to-sound-synthesis_amd/csrc/gemm_f16x2_ps.hip#$SCRATCH/gemm_f16x2_ps.hip#" "$ROOT/tools/probe/probe_ceiling.hip" > "$SCRATCH/probe_ceiling.hip"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I "$ROOT/include" -I "$SCRATCH" "$SCRATCH/probe_ceiling.hip" -lrocm_smi64 \
    -o "$ROOT/tools/probe/probe_ceiling"
rm -rf "$SCRATCH"
echo "$ROOT/tools/probe/probe_ceiling"
