#!/bin/bash
# Probe builds of the product library with extra defines on ONE source file:
#   tools/build_variant.sh <name> <file.hip> <defines...>  ->  gpurun_ab_<name>.so
# (git-ignored; travels to the GPU box; selected with DIFFSOUND_LIB).  tools/build_ps_variant.sh = the same for gemm_f16x2_ps.hip.
set -eu
NAME=$1; SRC=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
python "$ROOT/text-to-sound-synthesis_amd/build.py" > /dev/null
OBJ=$ROOT/text-to-sound-synthesis_amd/csrc/obj
BASE=$(basename "$SRC" .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -I "$ROOT/include" -I "$ROOT/text-to-sound-synthesis_amd/csrc" \
    -c "$ROOT/text-to-sound-synthesis_amd/csrc/$BASE.hip" -o "/tmp/${BASE}_$NAME.o"
OBJS=$(ls "$OBJ"/*.o | grep -v "/$BASE.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/gpurun_ab_$NAME.so" $OBJS "/tmp/${BASE}_$NAME.o"
echo "$ROOT/gpurun_ab_$NAME.so"
