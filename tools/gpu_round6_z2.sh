#!/bin/bash
# Round 6: after the small-launch column sums, the AdamW pass with a whole chunk's loads in flight, and the 128 x 128 rule for
# 193..256 tiles -- training kernel tests, the kernel profile of the captured iteration, and its rate.
O=gpurun_out/${1:-r06z2}
mkdir -p $O
python -m pytest tests/test_hip_train_kernels.py tests/test_hip_train_batch.py tests/test_hip_split_gemm.py -m gpu -x -q 2>&1 | tail -3 | tee $O/tests.txt
bash tools/gpu_round6_q.sh ${1:-r06z2} > /dev/null 2>&1
head -30 $O/train_graph_kernel_top.txt | cut -c1-175; tail -8 $O/train_graph_kernel_top.txt | cut -c1-150
python tools/bench_train.py --graph --steps 200 --prefetch > $O/bench_train.json 2> $O/bench_train.err; python - "$O/bench_train.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("it_per_s_sustained", "it_per_s_replay", "ms_per_replay_median", "recaptures", "loss", "grad_norm")})
PY
