#!/bin/bash
# Round 6: leading whole rounds of samples on the per-sample program + the rest as a second launch (N = 4096 layers of the
# training step): bit-identity test, the two shapes' time, the training tests and the iteration's rate.
O=gpurun_out/${1:-r06pt}
mkdir -p $O
python -m pytest tests/test_hip_widening.py tests/test_hip_train_batch.py tests/test_hip_train_kernels.py -m gpu -x -q 2>&1 | tail -3 | tee $O/tests.txt
python tools/train_gemm_ab.py --per-sample 2>&1 | grep "t9" | tee $O/per_sample_auto.txt
python tools/bench_train.py --graph --steps 200 --prefetch > $O/bench_train.json 2> $O/bench_train.err; python - "$O/bench_train.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("it_per_s_sustained", "it_per_s_replay", "ms_per_replay_median", "recaptures", "loss", "grad_norm")})
PY
