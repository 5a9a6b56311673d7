#!/bin/bash
# Round-3 measurement set (run from the repo root on the GPU box): bench line, kernel-trace stats, two PMC passes, the
# other configs of BASELINE.json.  Outputs under gpurun_out/final; the summaries are copied into profiles/ afterwards.
set -o pipefail
R=$(pwd); mkdir -p gpurun_out/final && cd /tmp && export TMPDIR=/tmp && cd $R
timeout -k 10 600 python3 bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err || exit 1
echo bench done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/kt -o kt -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/final/kt.log 2>&1 || exit 2
echo kernel trace done
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/final/pmc_fetch -o f -- python3 bench.py --pmc-workload > gpurun_out/final/pmc_f.log 2>&1 || exit 3
echo fetch done
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/final/pmc_write -o w -- python3 bench.py --pmc-workload > gpurun_out/final/pmc_w.log 2>&1 || exit 4
echo write done
python3 profiles/trace_window.py $(find gpurun_out/final/kt -name "*kernel_trace.csv" | head -1) > gpurun_out/final/roofline_pass_kernel_stats.csv || exit 5
cp $(find gpurun_out/final/kt -name "*kernel_stats.csv" | head -1) gpurun_out/final/bench_kernel_stats.csv
python3 profiles/pmc_traffic.py $(find gpurun_out/final/pmc_fetch -name "*counter_collection.csv" | head -1) $(find gpurun_out/final/pmc_write -name "*counter_collection.csv" | head -1) 768 gpurun_out/final/traffic.json || exit 6
rm -rf gpurun_out/final/kt gpurun_out/final/pmc_fetch gpurun_out/final/pmc_write
echo "other configs"
timeout -k 10 500 python3 bench.py --batch 2 --no-cpu-baseline > gpurun_out/final/bench_b2.json 2>> gpurun_out/final/bench.err || exit 7
timeout -k 10 500 python3 bench.py --res 1024 --batch 4 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/final/bench_1024_b4.json 2>> gpurun_out/final/bench.err || exit 8
timeout -k 10 300 python3 bench.py --mode decode --batch 1 --no-cpu-baseline > gpurun_out/final/bench_decode_b1.json 2>> gpurun_out/final/bench.err || exit 9
timeout -k 10 300 python3 bench.py --mode decode --batch 32 --no-cpu-baseline > gpurun_out/final/bench_decode_b32.json 2>> gpurun_out/final/bench.err || exit 10
ls -la gpurun_out/final
