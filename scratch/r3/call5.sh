set -e
mkdir -p gpurun_out/r3
timeout -k 10 400 python -u -m pytest tests/test_gpu_round3.py -k "two_group" -x -q --timeout 120 --timeout-method thread > gpurun_out/r3/tests_pp.log 2>&1 || { tail -60 gpurun_out/r3/tests_pp.log; exit 1; }
tail -2 gpurun_out/r3/tests_pp.log
OPT=attn_pp VALS=-1,1 timeout -k 10 300 python -u scratch/r3/attn_ab.py 1:5:9216:9216 2:5:9216:9216 1:10:2304:2304 2:10:2304:2304 4:5:16384:16384 2>&1 | tee gpurun_out/r3/attn_pp_ab.txt
echo "== row sums by v_add_f32 in attn64x2 (A) and attn64pp (B)"
SDMI_LIB_PATH=$PWD/scratch/r3/libadd.so OPT=attn_pp VALS=-1,1 timeout -k 10 300 python -u scratch/r3/attn_ab.py 1:5:9216:9216 2:5:9216:9216 2:10:2304:2304 2>&1 | tee gpurun_out/r3/attn_add_ab.txt
