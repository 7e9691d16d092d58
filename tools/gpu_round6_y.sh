#!/bin/bash
# Round 6: the new tile rule of the packed-operand split GEMM (96 x 128 tile; 128 x 128 from 342 tiles on) -- GEMM / training
# tests, the batch sweep again ('auto' should now sit on the best column), the training iteration's rate.
O=gpurun_out/${1:-r06y}
mkdir -p $O
python -m pytest tests/test_hip_split_gemm.py tests/test_hip_train_kernels.py tests/test_hip_train_batch.py -m gpu -x -q 2>&1 | tail -4 | tee $O/tests.txt
python tools/train_gemm_ab.py --batch-sweep 2>&1 | grep "^B=" > $O/batch_sweep_new_rule.txt; cut -c1-150 $O/batch_sweep_new_rule.txt | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$(NF-9),$(NF-8),$(NF-7),$(NF-6),$(NF-5)}'
python tools/bench_train.py --graph --steps 200 --prefetch > $O/bench_train.json 2> $O/bench_train.err; python - "$O/bench_train.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("it_per_s_sustained", "it_per_s_replay", "ms_per_replay_median", "recaptures", "loss", "grad_norm")})
PY
python tools/bench_train.py --graph --steps 100 --from-tokens 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('from tokens:', {k:d[k] for k in ('it_per_s_sustained','it_per_s_replay','recaptures')})"
