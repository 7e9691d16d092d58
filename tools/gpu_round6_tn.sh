#!/bin/bash
# Round 6: the TN mode of the split GEMM (dW from the row forms through transposing LDS reads): bit-identity test, then its
# speed against the transposed-planes route on the training step's dW shapes.
O=gpurun_out/${1:-r06tn}
mkdir -p $O
python -m pytest tests/test_hip_train_kernels.py -m gpu -x -q -k "tn_from_row or split_k_groups" 2>&1 | tail -15 | tee $O/tests.txt
python tools/train_gemm_ab.py --tn 2>&1 | grep "^dW" | tee $O/tn_vs_transposed.txt
