#!/bin/bash
# round 5: A/B of the paired dX + dW launch (tools/bench_train.py --pair-ranges 0 | 2 | 4), after the kernel tests
set -x
O=gpurun_out/${1:-r05u}
mkdir -p $O
timeout 600 python -m pytest tests/test_hip_train_kernels.py -m gpu -q 2>&1 | tail -8 > $O/pytest_train.log
tail -4 $O/pytest_train.log
for r in 0 2 4 0 4; do
  timeout 200 python tools/bench_train.py --precision f16x2 --graph --steps 20 --warmup 3 --pair-ranges $r > $O/bench_train_pair$r.json 2>> $O/bench_train.err
  echo "pair_ranges $r: $(grep -o '"value": [0-9.]*' $O/bench_train_pair$r.json)" | tee -a $O/pair_ab.txt
done
tail -3 $O/bench_train.err
