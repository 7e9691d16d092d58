#!/bin/bash
# Round 6, first GPU call: the new parity tests (training iteration from the reference's batch at 19 layers / B = 20 on both
# weight profiles, trained-like chain, full-length waveforms, the attention-backward monitor) and the sustained training rate.
O=gpurun_out/r06a
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_hip_train_batch.py -m gpu -x -q -s > $O/train_batch_tests.log 2>&1; echo "train_batch rc=$?" | tee -a $O/rc.txt
tail -25 $O/train_batch_tests.log
timeout 600 python -m pytest tests/test_hip_train_kernels.py -m gpu -x -q > $O/train_kernel_tests.log 2>&1; echo "train_kernels rc=$?" | tee -a $O/rc.txt
tail -5 $O/train_kernel_tests.log
timeout 900 python -m pytest tests/test_hip_full_config_parity.py -m gpu -q > $O/full_config_tests.log 2>&1; echo "full_config rc=$?" | tee -a $O/rc.txt
tail -30 $O/full_config_tests.log
for mode in "" "--from-tokens"; do
  timeout 300 python tools/bench_train.py --graph --steps 200 --warmup 5 $mode > $O/bench_train_200$mode.json 2> $O/bench_train_200$mode.err
  echo "bench_train $mode rc=$?" | tee -a $O/rc.txt
  cut -c1-1500 $O/bench_train_200$mode.json; tail -3 $O/bench_train_200$mode.err
done
cp gpurun_out/n1_parity_*.json $O/ 2>/dev/null
