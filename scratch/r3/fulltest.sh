set -e
mkdir -p gpurun_out/r3
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3/tests_d.log 2>&1 || { tail -60 gpurun_out/r3/tests_d.log; exit 1; }
tail -3 gpurun_out/r3/tests_d.log
