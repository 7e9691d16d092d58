#!/bin/bash
# round 5, last call: the GPU suite and smoke() on the final tree; the saturation monitor over 100 graphed training iterations (window
# upper bound 2^15 as shipped, and 2^16); bench.py with its defaults
set -x
O=gpurun_out/${1:-r05last}
mkdir -p $O
timeout 400 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/gpu_suite.log
tail -3 $O/gpu_suite.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for hi in 15 16; do
  timeout 150 python tools/bench_train.py --precision f16x2 --graph --steps 100 --warmup 3 --monitor-hi $hi > $O/bench_train_100_hi$hi.json 2>> $O/bench_train.err
  python - <<PY
import json
d = json.load(open("$O/bench_train_100_hi$hi.json"))
print("monitor hi $hi:", {k: d[k] for k in ("value", "recaptures", "monitor_log2", "loss_scale_exp", "loss", "grad_norm")})
PY
done 2>&1 | tee $O/monitor_ab.txt
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench.err
cut -c1-600 $O/bench_default.json
