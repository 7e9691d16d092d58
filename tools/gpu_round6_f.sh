#!/bin/bash
# Round 6: the whole GPU suite and smoke() on the tree, the round's profiling recipe (PMC passes, bench.py default line incl. the
# 200-iteration training leg and the CPU baseline, the same under rocprofv3 --stats), and bench.py with the driver's arguments.
O=gpurun_out/${1:-r06f}
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q > $O/gpu_suite.log 2>&1; echo "gpu_suite rc=$?" | tee -a $O/rc.txt
tail -45 $O/gpu_suite.log | cut -c1-600
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt; tail -1 $O/smoke.log | cut -c1-400
bash tools/profile_round.sh ${1:-r06f} > $O/profile_round.log 2>&1; echo "profile_round rc=$?" | tee -a $O/rc.txt
cut -c1-2500 $O/${1:-r06f}_bench_default.json
timeout 600 python bench.py --steps 20 --warmup 2 --no-cpu-baseline > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "bench20 rc=$?" | tee -a $O/rc.txt
cut -c1-700 $O/bench_steps20.json
