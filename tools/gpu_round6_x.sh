#!/bin/bash
# Round 6: the 96 x 128 tile (four waves side by side) of the packed-operand split GEMM -- bit-identity tests, then the training
# shapes' sweep (tools/train_gemm_ab.py: tile 3 next to 0 / 1 / 2).
O=gpurun_out/${1:-r06x}
mkdir -p $O
python -m pytest tests/test_hip_split_gemm.py tests/test_hip_train_kernels.py -m gpu -x -q 2>&1 | tail -4 | tee $O/tests.txt
python tools/train_gemm_ab.py 2>&1 | grep -v amdgpu.ids | tee $O/train_gemm_tile_sweep.txt | cut -c1-260
