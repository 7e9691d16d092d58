#!/bin/bash
# round 5, training-step check on the GPU box: kernel tests of the packed-operand path, the graphed iteration's rate, per-kernel times of the eager iteration
set -x
O=gpurun_out/${1:-r05c}
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_train_kernels.py -m gpu -q 2>&1 | tail -30 > $O/pytest_train.log
tail -8 $O/pytest_train.log
timeout 300 python tools/bench_train.py --precision f16x2 --graph --steps 10 --warmup 3 > $O/bench_train_graph.json 2> $O/bench_train_graph.err
cut -c1-400 $O/bench_train_graph.json; tail -3 $O/bench_train_graph.err
bash tools/train_profile.sh $O 2>&1 | tail -45 > $O/train_kernel_top.txt
cat $O/train_kernel_top.txt
