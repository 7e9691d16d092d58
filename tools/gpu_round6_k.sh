#!/bin/bash
# Round 6: long sustained training runs -- 1000 iterations of new batches on init-like weights (the switch to importance sampling
# of the timesteps happens inside), 300 on trained-like weights (heavy-tailed gradients): rate, re-captures and their reasons,
# monitor readings.
O=gpurun_out/${1:-r06k}
mkdir -p $O
export PYTHONUNBUFFERED=1
for v in "init1000 --steps 1000 --prefetch" "trained300 --steps 300 --prefetch --weights trained"; do
  set -- $v; name=$1; shift
  timeout 600 python tools/bench_train.py --graph --warmup 5 "$@" > $O/bench_train_$name.json 2> $O/bench_train_$name.err
  python - <<PY
import json
try:
    d = json.load(open("$O/bench_train_$name.json"))
    print("%-12s" % "$name", {k: (round(d[k], 3) if isinstance(d[k], float) else d[k]) for k in ("steps", "it_per_s_sustained", "it_per_s_replay", "recaptures", "recapture_reasons", "monitor_log2", "loss_scale_exp", "loss", "grad_norm", "weights")})
except Exception as e:
    print("$name failed:", e); print(open("$O/bench_train_$name.err").read()[-2500:])
PY
done 2>&1 | tee $O/bench_train_long.txt
