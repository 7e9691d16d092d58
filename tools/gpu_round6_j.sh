#!/bin/bash
# Round 6: the whole GPU suite, smoke(), and bench.py with the driver's arguments on the tree.
O=gpurun_out/${1:-r06j}
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -m gpu -q > $O/gpu_suite.log 2>&1; echo "gpu_suite rc=$?" | tee -a $O/rc.txt
tail -42 $O/gpu_suite.log | cut -c1-900
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt; tail -1 $O/smoke.log | cut -c1-400
timeout 900 python bench.py --steps 20 --warmup 2 > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "bench20 rc=$?" | tee -a $O/rc.txt
cut -c1-3500 $O/bench_steps20.json
