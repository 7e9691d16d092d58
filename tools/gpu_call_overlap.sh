set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02_ov
timeout 600 python -m pytest tests/test_hip_train_kernels.py -m gpu -q -s -p no:cacheprovider -k "overlapped or graphed or vs_oracle" > gpurun_out/r02_ov/tests.log 2>&1; tail -4 gpurun_out/r02_ov/tests.log
for MODE in "f16x2" "f16x2 --overlap-dw" "f16x2 --graph" "f16x2 --graph --overlap-dw" "fp32 --overlap-dw"; do
  NAME=$(echo "$MODE" | tr -d ' -')
  timeout 400 python tools/bench_train.py --precision $MODE --steps 6 --warmup 2 > gpurun_out/r02_ov/bt_$NAME.json 2> gpurun_out/r02_ov/bt_$NAME.err
  echo "$MODE rc=$? $(python -c "import json;d=json.load(open('gpurun_out/r02_ov/bt_$NAME.json'));print(round(d['value'],2),'it/s',round(d['ms_per_step'],1),'ms loss',round(d['loss'],3))" 2>&1 | tail -1)"; grep -v amdgpu.ids gpurun_out/r02_ov/bt_$NAME.err | tail -2
done
