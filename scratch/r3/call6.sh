set -e
mkdir -p gpurun_out/r3
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3/tests_c.log 2>&1 || { tail -60 gpurun_out/r3/tests_c.log; exit 1; }
tail -3 gpurun_out/r3/tests_c.log
timeout -k 10 300 python bench.py --steps 2 --warmup 1 > gpurun_out/r3/bench_c.json 2> gpurun_out/r3/bench_c.err || { tail -20 gpurun_out/r3/bench_c.err; exit 1; }
cat gpurun_out/r3/bench_c.json | cut -c1-600
