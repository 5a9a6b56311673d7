#!/bin/bash
# GPU call: new conv / GEMM loader tests, A/B timing, conv PMC
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 400 python -u -m pytest tests/test_gpu_round3.py tests/test_gpu_kernels.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r3/tests_b.log 2>&1
rc=$?; tail -15 gpurun_out/r3/tests_b.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python scratch/r3/conv_ab.py > gpurun_out/r3/conv_ab.txt 2>&1 || { tail -20 gpurun_out/r3/conv_ab.txt; exit 1; }
cat gpurun_out/r3/conv_ab.txt
timeout -k 10 200 python scratch/r3/gemm_ab.py > gpurun_out/r3/gemm_ab.txt 2>&1 || { tail -20 gpurun_out/r3/gemm_ab.txt; exit 1; }
cat gpurun_out/r3/gemm_ab.txt
bash scratch/r3/pmc_op.sh gpurun_out/r3/pmc_conv_v2.txt conv_win conv:96:640:320:2 conv:48:640:640:2 conv:96:320:320:1 || exit 1
grep -E "BANK_CONFLICT|IDX_ACTIVE|INSTS_(MFMA|VALU |SALU)|GRBM|COEXEC|MFMA_BUSY" gpurun_out/r3/pmc_conv_v2.txt
