#!/bin/bash
# Round 6, GPU call D: the per-wave dS normalisation of the attention backward, gradient parity at 19 layers / B = 20 on both
# weight profiles, the sustained training rate, and the price of splitting the long-contraction forward GEMMs.
O=gpurun_out/${1:-r06d}
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_hip_train_kernels.py -m gpu -q -s -k "attention_backward" > $O/attn_bwd_tests.log 2>&1; echo "attn_bwd rc=$?" | tee -a $O/rc.txt
grep -n "rel err\|monitor\|passed\|failed\|Error" $O/attn_bwd_tests.log | cut -c1-330 | tail -30
timeout 1200 python -m pytest tests/test_hip_train_batch.py -m gpu -q -s > $O/train_batch_tests.log 2>&1; echo "train_batch rc=$?" | tee -a $O/rc.txt
grep -v "^  File\|^    " $O/train_batch_tests.log | grep -n "grad-norm\|site exp\|train L19\|train batch\|passed\|failed\|^E  " | cut -c1-700 | tail -40
timeout 600 python -m pytest tests/test_hip_train_kernels.py tests/test_hip_split_gemm.py tests/test_hip_rccl.py -m gpu -q > $O/train_kernel_tests.log 2>&1; echo "train_kernels rc=$?" | tee -a $O/rc.txt
tail -5 $O/train_kernel_tests.log
timeout 300 python tools/bench_train.py --graph --steps 200 --warmup 5 > $O/bench_train_200.json 2> $O/bench_train_200.err
echo "bench_train rc=$?" | tee -a $O/rc.txt
cut -c1-1300 $O/bench_train_200.json; tail -3 $O/bench_train_200.err
timeout 300 python tools/train_gemm_ab.py 2>&1 | grep -v amdgpu.ids > $O/train_gemm_ab.txt; tail -12 $O/train_gemm_ab.txt
