#!/bin/bash
# Probe builds of the product library with extra defines on ONE source file:
#   tools/build_variant.sh <name> <file.hip>[,<file2.hip>...] <defines...>  ->  gpurun_ab_<name>.so
# (git-ignored; travels to the GPU box; selected with DIFFSOUND_LIB).  tools/build_ps_variant.sh = the same for gemm_f16x2_ps.hip.
# The product sources carry no probe code: the in-kernel time stamps (-DPS_TIMING, -DAH_TIMING), the ablation switches
# (-DPS_ABLATE=bits) and the tuning alternatives of the attention kernel (-DAH_TPC=.. etc.) live in tools/probe/*.patch, which
# is applied here to a SCRATCH copy of csrc/ before the one file is compiled (tools/probe/README.md).
set -eu
NAME=$1; SRC=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
python "$ROOT/text-to-sound-synthesis_amd/build.py" > /dev/null
OBJ=$ROOT/text-to-sound-synthesis_amd/csrc/obj
SCRATCH=$(mktemp -d /tmp/ds_probe_src.XXXXXX)
cp "$ROOT"/text-to-sound-synthesis_amd/csrc/*.hip "$ROOT"/text-to-sound-synthesis_amd/csrc/*.inc "$ROOT"/text-to-sound-synthesis_amd/csrc/*.h "$SCRATCH"/
# PROBE_PATCHES: which patches (names under tools/probe/, space separated); default = the two that carry the -D switches
for P in ${PROBE_PATCHES:-ps_probe.patch attn_probe.patch}; do patch -s -d "$SCRATCH" -p1 < "$ROOT/tools/probe/$P"; done
OBJS=$(ls "$OBJ"/*.o)
NEW=""
for F in $(echo "$SRC" | tr ',' ' '); do
    BASE=$(basename "$F" .hip)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -I "$ROOT/include" -I "$SCRATCH" \
        -Rpass-analysis=kernel-resource-usage -c "$SCRATCH/$BASE.hip" -o "/tmp/${BASE}_$NAME.o" 2> "/tmp/${BASE}_$NAME.remarks"
    grep -B8 "ScratchSize \[bytes/lane\]: [1-9]" "/tmp/${BASE}_$NAME.remarks" | grep -o "Function Name: [^ ]*\|ScratchSize.*" | paste - - || true
    OBJS=$(echo "$OBJS" | grep -v "/$BASE.o")
    NEW="$NEW /tmp/${BASE}_$NAME.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/gpurun_ab_$NAME.so" $OBJS $NEW
rm -rf "$SCRATCH"
echo "$ROOT/gpurun_ab_$NAME.so"
