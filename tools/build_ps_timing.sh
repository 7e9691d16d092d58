#!/bin/bash
# Probe build: the product library with gemm_f16x2_ps.hip compiled -DPS_TIMING (in-kernel time stamps) -> gpurun_ab_timing.so
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
python "$ROOT/text-to-sound-synthesis_amd/build.py" > /dev/null
OBJ=$ROOT/text-to-sound-synthesis_amd/csrc/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPS_TIMING -I "$ROOT/include" -I "$ROOT/text-to-sound-synthesis_amd/csrc" \
    -c "$ROOT/text-to-sound-synthesis_amd/csrc/gemm_f16x2_ps.hip" -o /tmp/gemm_f16x2_ps_timing.o
OBJS=$(ls "$OBJ"/*.o | grep -v gemm_f16x2_ps.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/gpurun_ab_timing.so" $OBJS /tmp/gemm_f16x2_ps_timing.o
echo "$ROOT/gpurun_ab_timing.so"
