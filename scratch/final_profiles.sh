#!/bin/bash
# Round-end measurement set (run from the repo root on the GPU box): bench line, kernel-trace stats, two PMC passes.
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
find gpurun_out/final -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | head
# the raw traces are large: keep the summaries
python3 profiles/trace_window.py $(find gpurun_out/final/kt -name "*kernel_trace.csv" | head -1) > gpurun_out/final/roofline_pass_kernel_stats.csv || exit 5
head -5 gpurun_out/final/roofline_pass_kernel_stats.csv
find gpurun_out/final -name "*kernel_trace.csv" -delete
