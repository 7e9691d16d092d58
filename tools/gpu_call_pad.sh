set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02_pad
timeout 300 python -m pytest tests/test_hip_widening.py -m gpu -q -s -x -p no:cacheprovider -k "padded_row" > gpurun_out/r02_pad/pad_test.log 2>&1; grep -v amdgpu gpurun_out/r02_pad/pad_test.log | tail -6
for PAD in 1 0 1 0; do
  DIFFSOUND_PAD_ROWS=$PAD timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02_pad/bench_pad$PAD.json 2> gpurun_out/r02_pad/bench_pad$PAD.err
  python -c "import json;d=json.load(open('gpurun_out/r02_pad/bench_pad$PAD.json'));r=d['roofline'];print('PAD=$PAD', d['value'],'clips/s', d['ms_per_step'],'ms  frac',r['frac'],'avg_us',r['avg_launch_us'], r['kernel'][:40])"
done
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r02_pad/gpu_suite.log 2>&1; tail -3 gpurun_out/r02_pad/gpu_suite.log
