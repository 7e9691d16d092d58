#!/bin/bash
# Probe builds of the product library with extra defines on gemm_f16x2_ps.hip (e.g. -DPS_TIMING for tools/ps_timing.py):
#   tools/build_ps_variant.sh <name> <defines...>  ->  gpurun_ab_<name>.so      (= tools/build_variant.sh for that file)
set -eu
NAME=$1; shift
exec "$(dirname "$0")/build_variant.sh" "$NAME" gemm_f16x2_ps.hip "$@"
