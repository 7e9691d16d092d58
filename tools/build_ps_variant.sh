#!/bin/bash
# Probe builds of the product library with extra defines on gemm_f16x2_ps.hip:  tools/build_ps_variant.sh <name> <defines...>
#   -> gpurun_ab_<name>.so (git-ignored; travels to the GPU box; selected with DIFFSOUND_LIB)
set -eu
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
python "$ROOT/text-to-sound-synthesis_amd/build.py" > /dev/null
OBJ=$ROOT/text-to-sound-synthesis_amd/csrc/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -I "$ROOT/include" -I "$ROOT/text-to-sound-synthesis_amd/csrc" \
    -c "$ROOT/text-to-sound-synthesis_amd/csrc/gemm_f16x2_ps.hip" -o "/tmp/gemm_f16x2_ps_$NAME.o"
OBJS=$(ls "$OBJ"/*.o | grep -v gemm_f16x2_ps.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/gpurun_ab_$NAME.so" $OBJS "/tmp/gemm_f16x2_ps_$NAME.o"
echo "$ROOT/gpurun_ab_$NAME.so"
