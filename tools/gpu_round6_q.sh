#!/bin/bash
# Round 6: rocprofv3 --kernel-trace --stats over the captured training iteration from the reference's batch on the final tree
# (60 replays; prefetch on), per-kernel table + the stock-torch-kernel share.
O=gpurun_out/${1:-r06q}
mkdir -p $O
bash tools/train_profile.sh $O --graph --steps 60 --prefetch 2>&1 | tail -40 > $O/train_graph_kernel_top.txt
python - "$O/train_kernel_stats.csv" <<'PY' | tee -a $O/train_graph_kernel_top.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = 65.0                                          # 2 warm-up + 60 timed iterations + capture warm-up + 2 calibration passes
stock = [(r["Name"], int(r["Calls"]), float(r["TotalDurationNs"])) for r in rows if "at::native" in r["Name"] or "rocclr" in r["Name"]]
print("stock torch / runtime kernels: %.2f ms per iteration in %.0f launches (of %.1f ms of kernels per iteration)"
      % (sum(x[2] for x in stock) / 1e6 / n, sum(x[1] for x in stock) / n, sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / n))
for name, calls, ns in sorted(stock, key=lambda x: -x[2])[:6]:
    print("  %6.1f calls/it %7.3f ms/it  %s" % (calls / n, ns / 1e6 / n, name[:120]))
PY
cut -c1-180 $O/train_graph_kernel_top.txt | tail -50
