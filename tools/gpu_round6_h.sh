#!/bin/bash
# Round 6: (1) the training-parity file again, (2) the round's profiling recipe (PMC passes, default bench line, the same under
# rocprofv3 --stats), (3) bench.py with the driver's arguments, (4) the LN-fold timing probe (tools/gpu_round6_g.sh).
T=${1:-r06h}
O=gpurun_out/$T
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_hip_train_batch.py tests/test_hip_rccl.py -m gpu -q > $O/train_batch_tests.log 2>&1; echo "train_batch rc=$?" | tee -a $O/rc.txt
tail -3 $O/train_batch_tests.log
bash tools/profile_round.sh $T > $O/profile_round.log 2>&1; echo "profile_round rc=$?" | tee -a $O/rc.txt
cut -c1-3000 $O/${T}_bench_default.json
timeout 600 python bench.py --steps 20 --warmup 2 --no-cpu-baseline > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "bench20 rc=$?" | tee -a $O/rc.txt
cut -c1-400 $O/bench_steps20.json
bash tools/gpu_round6_g.sh $T 2>&1 | tee $O/ln_fold_probe.txt
du -sh $O
