#!/bin/bash
# Round 6: the multi-tensor EMA kernel -- its test, the training kernel / batch tests, and the sustained rate with it.
O=gpurun_out/${1:-r06n}
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_hip_train_kernels.py tests/test_hip_train_batch.py tests/test_hip_rccl.py -m gpu -q > $O/train_tests.log 2>&1; echo "train_tests rc=$?" | tee -a $O/rc.txt
tail -4 $O/train_tests.log | cut -c1-300
for v in "prefetch --prefetch" "inline"; do
  set -- $v; name=$1; shift
  timeout 300 python tools/bench_train.py --graph --steps 200 --warmup 5 "$@" > $O/bench_train_$name.json 2> $O/bench_train_$name.err
  python - <<PY
import json
try:
    d = json.load(open("$O/bench_train_$name.json"))
    print("%-10s" % "$name", {k: (round(d[k], 3) if isinstance(d[k], float) else d[k]) for k in ("it_per_s_sustained", "it_per_s_replay", "ms_per_iteration_max", "recaptures", "loss", "prefetch")})
except Exception as e:
    print("$name failed:", e); print(open("$O/bench_train_$name.err").read()[-2000:])
PY
done 2>&1 | tee $O/bench_train_ema.txt
