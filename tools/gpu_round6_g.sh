#!/bin/bash
# Round 6: LN-fold TIMING PROBE (tools/probe/ln_fold_probe.patch -> gpurun_ab_lnfold.so): the sampling step with the blocks' 57
# LayerNorm / AdaLN launches removed (mode 1) and with the residual GEMMs additionally writing scaled packed planes + row partial
# sums as the producer side of the fold would (mode 2) -- values are garbage, only the time counts.  Kill criterion of the fold
# (decided before this call): a step above 26.8 ms (27.7 now) even in this optimistic form.
O=gpurun_out/${1:-r06g}
mkdir -p $O
export PYTHONUNBUFFERED=1
run() {   # name, lib, mode
  DIFFSOUND_LIB=$2 DS_LNFOLD_PROBE=$3 timeout 300 python bench.py --no-train-leg --no-cpu-baseline --steps 4 --warmup 2 > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
d = json.load(open("$O/bench_$1.json"))
r = d.get("roofline") or {}
print("%-28s ms_per_step %8.2f  clips/s %6.3f  sample %s  GEMM avg %s us frac %s" % ("$1", d["ms_per_step"], d["value"], (d.get("stage_ms") or {}).get("sample"), r.get("avg_launch_us"), r.get("frac")))
PY
}
run product "" 0
run probe_mode0 $PWD/gpurun_ab_lnfold.so 0
run probe_mode1_no_ln $PWD/gpurun_ab_lnfold.so 1
run probe_mode2_no_ln_producer_planes $PWD/gpurun_ab_lnfold.so 2
run product_again "" 0
