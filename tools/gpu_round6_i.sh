#!/bin/bash
# Round 6: the training parity file with the per-variant report, and the sustained training rate with / without the next batch's
# prologue prefetched on a side stream, at calibration targets 2^6 / 2^8 / 2^10 (monitor readings, re-captures).
O=gpurun_out/${1:-r06i}
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_hip_train_batch.py -m gpu -q > $O/train_batch_tests.log 2>&1; echo "train_batch rc=$?" | tee -a $O/rc.txt
grep "train L19\|train batch\|passed\|failed" $O/train_batch_tests.log | cut -c1-1200
for v in "base" "prefetch --prefetch" "t8 --calib-log2 8" "t10 --calib-log2 10" "t8_prefetch --calib-log2 8 --prefetch"; do
  set -- $v; name=$1; shift
  timeout 300 python tools/bench_train.py --graph --steps 200 --warmup 5 "$@" > $O/bench_train_$name.json 2> $O/bench_train_$name.err
  python - <<PY
import json
try:
    d = json.load(open("$O/bench_train_$name.json"))
    print("%-12s" % "$name", {k: (round(d[k], 3) if isinstance(d[k], float) else d[k]) for k in ("it_per_s_sustained", "it_per_s_replay", "recaptures", "monitor_log2", "loss_scale_exp", "loss", "prefetch")})
except Exception as e:
    print("$name failed:", e); print(open("$O/bench_train_$name.err").read()[-1500:])
PY
done 2>&1 | tee $O/bench_train_ab.txt
