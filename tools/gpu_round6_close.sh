#!/bin/bash
# Round 6, closing evidence on the final tree: the round's profiling recipe (PMC passes, bench default, the same command under
# rocprofv3 --kernel-trace --stats) and the two single-GPU side legs of BASELINE configs[1] / configs[3].
O=gpurun_out/${1:-r06z9}
mkdir -p $O
bash tools/profile_round.sh ${1:-r06z9} > $O/profile_round.log 2>&1
python bench.py --transformer-only --batch 32 --steps 2 --warmup 1 --no-train-leg --no-cpu-baseline > $O/bench_cfg1_b32_transformer_only.json 2> $O/cfg1.err
cut -c1-200 $O/bench_cfg1_b32_transformer_only.json
python bench.py --codes 512 --steps 2 --warmup 1 --no-train-leg --no-cpu-baseline > $O/bench_cfg3_k512_b64_1gpu.json 2> $O/cfg3.err
cut -c1-200 $O/bench_cfg3_k512_b64_1gpu.json
ls -la $O | head -30
cut -c1-400 $O/${1:-r06z9}_bench_default.json
