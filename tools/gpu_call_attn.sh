set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02_attn
DIFFSOUND_LIB=$PWD/gpurun_ab_attn_timing.so timeout 200 python tools/attn_timing.py 272 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_attn/attn_timeline_after.txt
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "atten or full_config or models or padded" > gpurun_out/r02_attn/tests.log 2>&1; tail -3 gpurun_out/r02_attn/tests.log
for V in new old new old; do
  if [ $V = old ]; then export DIFFSOUND_LIB=$PWD/gpurun_ab_oldattn.so; else unset DIFFSOUND_LIB; fi
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r02_attn/bench_$V.json 2> gpurun_out/r02_attn/bench_$V.err
  python -c "import json;d=json.load(open('gpurun_out/r02_attn/bench_$V.json'));print('$V', d['value'],'clips/s', d['ms_per_step'],'ms')"
done
