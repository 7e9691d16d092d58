#!/bin/bash
# Round 6, second GPU call: gradient parity at 19 layers / B = 20 with the per-site gradient scales, the pack kernel's new
# column-sum semantics, the sustained training rate again, and a per-kernel profile of the captured iteration (60 replays, so
# that capture-time kernels do not pollute the per-replay counts).
O=gpurun_out/${1:-r06b}
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_hip_train_batch.py -m gpu -q -s > $O/train_batch_tests.log 2>&1; echo "train_batch rc=$?" | tee -a $O/rc.txt
grep -v "^  File\|^    " $O/train_batch_tests.log | tail -40
timeout 600 python -m pytest tests/test_hip_train_kernels.py tests/test_hip_split_gemm.py -m gpu -x -q > $O/train_kernel_tests.log 2>&1; echo "train_kernels rc=$?" | tee -a $O/rc.txt
tail -5 $O/train_kernel_tests.log
timeout 300 python tools/bench_train.py --graph --steps 200 --warmup 5 > $O/bench_train_200.json 2> $O/bench_train_200.err
echo "bench_train rc=$?" | tee -a $O/rc.txt
cut -c1-1300 $O/bench_train_200.json; tail -3 $O/bench_train_200.err
true
true
