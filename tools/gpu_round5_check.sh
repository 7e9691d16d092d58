set -x
mkdir -p gpurun_out/r05a
python -m pytest tests -m gpu -q -x 2>&1 | tail -40 > gpurun_out/r05a/pytest.log
python bench.py --steps 3 --warmup 1 > gpurun_out/r05a/bench.json 2> gpurun_out/r05a/bench.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05a/smoke.log 2>&1
tail -5 gpurun_out/r05a/pytest.log; cut -c1-1500 gpurun_out/r05a/bench.json; tail -3 gpurun_out/r05a/bench.err; tail -2 gpurun_out/r05a/smoke.log
