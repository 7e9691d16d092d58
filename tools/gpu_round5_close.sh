#!/bin/bash
# round-5 closing measurements on the final tree (run ON THE GPU BOX): the GPU suite, smoke(), bench.py with the driver's arguments, the
# two single-GPU side lines (BASELINE configs[1] and the K = 512 leg of configs[3])
set -x
O=gpurun_out/${1:-r05y}
mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/gpu_suite.log
tail -5 $O/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err
cut -c1-700 $O/bench_steps20.json
python bench.py --transformer-only --batch 32 --steps 2 --warmup 1 --no-train-leg --no-cpu-baseline > $O/bench_cfg1_b32_transformer_only.json 2> $O/cfg1.err
cut -c1-200 $O/bench_cfg1_b32_transformer_only.json
python bench.py --codes 512 --steps 2 --warmup 1 --no-train-leg --no-cpu-baseline > $O/bench_cfg3_k512_b64_1gpu.json 2> $O/cfg3.err
cut -c1-200 $O/bench_cfg3_k512_b64_1gpu.json
